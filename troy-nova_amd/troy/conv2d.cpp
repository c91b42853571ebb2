// conv2d.cpp -- see conv2d.h.  Host logic only; all arithmetic goes through the mirror's Evaluator (GPU).
#include "conv2d.h"

namespace troy { namespace linear {

static size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

Conv2dHelper::Conv2dHelper(size_t batch_size, size_t input_channels, size_t output_channels, size_t image_height, size_t image_width, size_t kernel_height,
                           size_t kernel_width, size_t poly_degree, MatmulObjective objective, MemoryPoolHandle pool)
    : batch_size(batch_size), input_channels(input_channels), output_channels(output_channels), image_height(image_height), image_width(image_width),
      kernel_height(kernel_height), kernel_width(kernel_width), slot_count(poly_degree), objective(objective), pool(std::move(pool)) {
    if (kernel_height == 0 || kernel_width == 0 || kernel_height > image_height || kernel_width > image_width)
        throw std::invalid_argument("[Conv2dHelper::Conv2dHelper] the kernel must fit inside the image.");
    determine_block();
}

void Conv2dHelper::determine_block() {
    // app/conv2d.cu:31-94: exhaustive search over (b, h, w, co) with ci filling the rest of the polynomial; the cost is the number of
    // ciphertexts that travel (which operands are encrypted depends on the objective)
    size_t best = static_cast<size_t>(-1);
    for (size_t b = batch_size; b >= 1; b--) {
        for (size_t h = std::min(image_height, slot_count / b); h >= kernel_height; h--) {
            for (size_t w = std::min(image_width, slot_count / b / h); w >= kernel_width; w--) {
                for (size_t co = std::min(output_channels, slot_count / b / h / w); co >= 1; co--) {
                    const size_t ci = std::min(slot_count / b / h / w / co, input_channels);
                    if (ci == 0) continue;
                    const size_t tiles = ceil_div(batch_size, b) * ceil_div(image_height - kernel_height + 1, h - kernel_height + 1) *
                                         ceil_div(image_width - kernel_width + 1, w - kernel_width + 1);
                    const size_t in_ct = tiles * ceil_div(input_channels, ci), out_ct = tiles * ceil_div(output_channels, co);
                    const size_t w_ct = ceil_div(input_channels, ci) * ceil_div(output_channels, co);
                    const size_t cost = objective == MatmulObjective::EncryptLeft ? in_ct + out_ct : objective == MatmulObjective::EncryptRight ? w_ct + out_ct : in_ct + out_ct + w_ct;
                    if (cost < best) { best = cost; batch_block = b; image_height_block = h; image_width_block = w; input_channel_block = ci; output_channel_block = co; }
                }
            }
        }
    }
    if (best == static_cast<size_t>(-1)) throw std::invalid_argument("[Conv2dHelper::determine_block] no valid blocking for these dimensions.");
}

size_t Conv2dHelper::get_total_batch_size() const {
    const size_t kh = kernel_height - 1, kw = kernel_width - 1;
    return ceil_div(batch_size, batch_block) * ceil_div(image_height - kh, image_height_block - kh) * ceil_div(image_width - kw, image_width_block - kw);
}

static void check_below(uint64_t v, uint64_t t) { if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus"); }

std::vector<uint64_t> Conv2dHelper::pack_weights(uint64_t t, const uint64_t* weights, size_t& rows, size_t& cols, size_t& len) const {
    return pack_weights_of<uint64_t>(weights, rows, cols, len, [t](uint64_t v) { check_below(v, t); });
}

std::vector<uint64_t> Conv2dHelper::pack_inputs(uint64_t t, const uint64_t* inputs, size_t& rows, size_t& cols, size_t& len) const {
    return pack_inputs_of<uint64_t>(inputs, rows, cols, len, [t](uint64_t v) { check_below(v, t); });
}

static uint64_t plain_modulus_of(const BatchEncoder& encoder) { return encoder.context()->first_context_data().value()->parms().plain_modulus().value(); }

Plain2d Conv2dHelper::encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_weights(plain_modulus_of(encoder), weights, rows, cols, len);
    return detail::encode_blocks_for_plain(encoder, packed, rows, cols, len, pool);
}
Cipher2d Conv2dHelper::encrypt_weights_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* weights) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_weights(plain_modulus_of(encoder), weights, rows, cols, len);
    return detail::encrypt_blocks(encryptor, encoder, packed, rows, cols, len, pool);
}
Plain2d Conv2dHelper::encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_inputs(plain_modulus_of(encoder), inputs, rows, cols, len);
    return detail::encode_blocks_for_plain(encoder, packed, rows, cols, len, pool);
}
Cipher2d Conv2dHelper::encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_inputs(plain_modulus_of(encoder), inputs, rows, cols, len);
    return detail::encrypt_blocks(encryptor, encoder, packed, rows, cols, len, pool);
}

Cipher2d Conv2dHelper::conv2d(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const {
    // app/conv2d.cu:356-404
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block), in_groups = ceil_div(input_channels, input_channel_block);
    if (a.size() != tiles) throw std::invalid_argument("[Conv2dHelper::conv2d] Input tile count incorrect.");
    if (w.size() != groups) throw std::invalid_argument("[Conv2dHelper::conv2d] Weight output-channel block count incorrect.");
    return detail::accumulate_products(evaluator, a[0][0], tiles, in_groups, groups,
                                       [&](size_t b, size_t i, size_t) { return &a[b][i]; }, [&](size_t, size_t i, size_t oc) { return &w[oc][i]; }, pool);
}

Cipher2d Conv2dHelper::conv2d_reverse(const Evaluator& evaluator, const Plain2d& a, const Cipher2d& w) const {
    // app/conv2d.cu:424-470
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block), in_groups = ceil_div(input_channels, input_channel_block);
    if (a.size() != tiles) throw std::invalid_argument("[Conv2dHelper::conv2d] Input tile count incorrect.");
    if (w.size() != groups) throw std::invalid_argument("[Conv2dHelper::conv2d] Weight output-channel block count incorrect.");
    return detail::accumulate_products(evaluator, w[0][0], tiles, in_groups, groups,
                                       [&](size_t, size_t i, size_t oc) { return &w[oc][i]; }, [&](size_t b, size_t i, size_t) { return &a[b][i]; }, pool);
}

Cipher2d Conv2dHelper::conv2d_cipher(const Evaluator& evaluator, const Cipher2d& a, const Cipher2d& w) const {
    // app/conv2d.cu:406-422 (CKKS / BGV: BFV operands in NTT form are refused by Evaluator::multiply, as in the reference)
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block);
    Cipher2d ret;
    for (size_t b = 0; b < tiles; b++) {
        std::vector<Ciphertext>& row = ret.new_row();
        for (size_t oc = 0; oc < groups; oc++) {
            Ciphertext acc;
            for (size_t i = 0; i < a[b].size(); i++) {
                Ciphertext prod;
                evaluator.multiply(a[b][i], w[oc][i], prod, pool);
                if (i == 0) acc = std::move(prod); else evaluator.add_inplace(acc, prod, pool);
            }
            row.push_back(std::move(acc));
        }
    }
    return ret;
}

Plain2d Conv2dHelper::encode_outputs_uint64s(const BatchEncoder& encoder, const uint64_t* outputs) const {
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block);
    std::vector<std::vector<uint64_t>> buffers(tiles * groups, std::vector<uint64_t>(slot_count, 0));
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { buffers[tile * groups + group][coefficient] = outputs[index]; });
    Plain2d out;
    for (size_t tile = 0; tile < tiles; tile++) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t g = 0; g < groups; g++) row.push_back(encoder.encode_polynomial_new(buffers[tile * groups + g], pool));
    }
    return out;
}

std::vector<uint64_t> Conv2dHelper::decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    (void)encoder;
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block), n = slot_count;
    std::vector<const Ciphertext*> all;
    for (const auto& r : outputs.data()) for (const Ciphertext& c : r) all.push_back(&c);
    if (all.size() != tiles * groups) throw std::invalid_argument("[Conv2dHelper::decrypt_outputs] Output ciphertext count incorrect");
    const std::vector<uint64_t> coeffs = decryptor.bfv_decrypt_to_host(all, pool);   // one batch, one copy back
    const size_t oyh = image_height - kernel_height + 1, oyw = image_width - kernel_width + 1;
    std::vector<uint64_t> out(batch_size * output_channels * oyh * oyw, 0);
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { out[index] = coeffs[(tile * groups + group) * n + coefficient]; });
    return out;
}

// ---- CKKS forms (app/conv2d.cu *_doubles): the same layouts through CKKSEncoder's polynomial encoding ------------------------------
static Plain2d encode_double_blocks(const CKKSEncoder& encoder, const std::vector<double>& packed, size_t rows, size_t cols, size_t len, std::optional<ParmsID> parms_id,
                                    double scale, MemoryPoolHandle pool) {
    Plain2d out;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t k = 0; k < cols; k++) {
            const auto begin = packed.begin() + static_cast<std::ptrdiff_t>((r * cols + k) * len);
            row.push_back(encoder.encode_float64_polynomial_new(std::vector<double>(begin, begin + static_cast<std::ptrdiff_t>(len)), parms_id, scale, pool));
        }
    }
    return out;
}

static Cipher2d encrypt_plain2d(const Encryptor& encryptor, const Plain2d& plain, MemoryPoolHandle pool) {
    Cipher2d out;
    for (const auto& prow : plain.data()) {
        std::vector<Ciphertext>& row = out.new_row();
        for (const Plaintext& p : prow) row.push_back(encryptor.encrypt_symmetric_new(p, true, pool));   // c1 travels as its seed
    }
    return out;
}

Plain2d Conv2dHelper::encode_weights_doubles(const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const {
    size_t rows, cols, len;
    const std::vector<double> packed = pack_weights_of<double>(weights, rows, cols, len, [](double) {});
    return encode_double_blocks(encoder, packed, rows, cols, len, parms_id, scale, pool);
}
Plain2d Conv2dHelper::encode_inputs_doubles(const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const {
    size_t rows, cols, len;
    const std::vector<double> packed = pack_inputs_of<double>(inputs, rows, cols, len, [](double) {});
    return encode_double_blocks(encoder, packed, rows, cols, len, parms_id, scale, pool);
}
Cipher2d Conv2dHelper::encrypt_weights_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const {
    return encrypt_plain2d(encryptor, encode_weights_doubles(encoder, weights, parms_id, scale), pool);
}
Cipher2d Conv2dHelper::encrypt_inputs_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const {
    return encrypt_plain2d(encryptor, encode_inputs_doubles(encoder, inputs, parms_id, scale), pool);
}

Plain2d Conv2dHelper::encode_outputs_doubles(const CKKSEncoder& encoder, const double* outputs, std::optional<ParmsID> parms_id, double scale) const {
    const size_t tiles = get_total_batch_size(), groups = (output_channels + output_channel_block - 1) / output_channel_block;
    std::vector<std::vector<double>> buffers(tiles * groups, std::vector<double>(slot_count, 0.0));
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { buffers[tile * groups + group][coefficient] = outputs[index]; });
    Plain2d out;
    for (size_t tile = 0; tile < tiles; tile++) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t g = 0; g < groups; g++) row.push_back(encoder.encode_float64_polynomial_new(buffers[tile * groups + g], parms_id, scale, pool));
    }
    return out;
}

std::vector<double> Conv2dHelper::decrypt_outputs_doubles(const CKKSEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    const size_t tiles = get_total_batch_size(), groups = (output_channels + output_channel_block - 1) / output_channel_block;
    std::vector<std::vector<double>> coeffs;
    for (const auto& r : outputs.data())
        for (const Ciphertext& ct : r) coeffs.push_back(encoder.decode_float64_polynomial_new(decryptor.decrypt_new(ct, pool), pool));
    if (coeffs.size() != tiles * groups) throw std::invalid_argument("[Conv2dHelper::decrypt_outputs] Output ciphertext count incorrect");
    const size_t oyh = image_height - kernel_height + 1, oyw = image_width - kernel_width + 1;
    std::vector<double> out(batch_size * output_channels * oyh * oyw, 0.0);
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { out[index] = coeffs[tile * groups + group][coefficient]; });
    return out;
}

std::vector<size_t> Conv2dHelper::output_terms() const {
    // app/conv2d.cu:478-492: every coefficient a full tile can carry (the same list for every ciphertext)
    const size_t interval = image_width_block * image_height_block;
    const size_t yh = image_height_block - kernel_height + 1, yw = image_width_block - kernel_width + 1;
    std::vector<size_t> required;
    for (size_t b = 0; b < batch_block; b++)
        for (size_t c = 0; c < output_channel_block; c++)
            for (size_t i = 0; i < yh; i++)
                for (size_t j = 0; j < yw; j++)
                    required.push_back((b * input_channel_block * output_channel_block + c * input_channel_block + input_channel_block - 1) * interval +
                                       (image_height_block - yh + i) * image_width_block + (image_width_block - yw + j));
    return required;
}

void Conv2dHelper::serialize_outputs(const Evaluator& evaluator, const Cipher2d& x, std::ostream& stream, CompressionMode mode) const {
    const std::vector<size_t> required = output_terms();
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block);
    if (x.size() != tiles) throw std::invalid_argument("[Conv2dHelper::serialize_outputs] Output ciphertext count incorrect");
    for (size_t b = 0; b < tiles; b++)
        for (size_t oc = 0; oc < groups; oc++) x[b][oc].save_terms(stream, evaluator.context(), required, pool, mode);
}

Cipher2d Conv2dHelper::deserialize_outputs(const Evaluator& evaluator, std::istream& stream) const {
    const std::vector<size_t> required = output_terms();
    const size_t tiles = get_total_batch_size(), groups = ceil_div(output_channels, output_channel_block);
    Cipher2d ret;
    for (size_t b = 0; b < tiles; b++) {
        std::vector<Ciphertext>& row = ret.new_row();
        for (size_t oc = 0; oc < groups; oc++) row.push_back(Ciphertext::load_terms_new(stream, evaluator.context(), required, pool));
    }
    return ret;
}

}}  // namespace troy::linear

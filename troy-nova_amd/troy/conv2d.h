// conv2d.h -- troy::linear::Conv2dHelper, the 2-d convolution of the reference's conv2d application (src/app/conv2d.{h,cu},
// examples/14_bfv_conv2d.cu) on top of the mirror API and the batched block helpers of matmul.h.
//
// Valid ("no padding") cross-correlation  y[b][co][i][j] = sum_{ci,ki,kj} x[b][ci][i+ki][j+kj] * w[co][ci][ki][kj]  as polynomial
// products: an image tile of (hb x wb) pixels per channel is laid out row-major in the coefficients, the kernel flipped, so that the
// wanted sums appear on fixed coefficients of the product.  Layout, as the reference (blk = hb*wb, cib/cob = channel blocks):
//   input  (batch b, channel ci, pixel (i, j) of the tile):   coefficient b*cib*cob*blk + ci*blk + i*wb + j
//   weight (co, ci, kernel (ki, kj)):                          coefficient co*cib*blk + (cib-1-ci)*blk + (kh-1-ki)*wb + (kw-1-kj)
//   output (b, co, (i, j)):                                    coefficient (b*cib*cob + co*cib + cib-1)*blk + (hb-yh+i)*wb + (wb-yw+j)
// Tiles overlap by (kh-1, kw-1) pixels so that every output pixel is produced by exactly one tile.
#pragma once
#include "matmul.h"

namespace troy { namespace linear {

class Conv2dHelper {
public:
    size_t batch_size, input_channels, output_channels, image_height, image_width, kernel_height, kernel_width, slot_count;
    size_t batch_block = 0, input_channel_block = 0, output_channel_block = 0, image_height_block = 0, image_width_block = 0;
    MatmulObjective objective;
    MemoryPoolHandle pool;

    Conv2dHelper(size_t batch_size, size_t input_channels, size_t output_channels, size_t image_height, size_t image_width, size_t kernel_height, size_t kernel_width,
                 size_t poly_degree, MatmulObjective objective = MatmulObjective::EncryptLeft, MemoryPoolHandle pool = MemoryPool::GlobalPool());

    // tiles: ceil(batch / bb) * tiles_h * tiles_w rows of the input / output Cipher2d
    size_t get_total_batch_size() const;

    // weights [co][ci][kh][kw] -> [ceil(co/cob)][ceil(ci/cib)]; inputs [b][ci][H][W] -> [tiles][ceil(ci/cib)]
    Plain2d encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const;
    Cipher2d encrypt_weights_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* weights) const;
    Plain2d encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const;
    Cipher2d encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const;
    // ret[tile][oc] = sum_i a[tile][i] * w[oc][i]  (one multiply_plain_accumulate launch)
    Cipher2d conv2d(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const;
    Cipher2d conv2d_reverse(const Evaluator& evaluator, const Plain2d& a, const Cipher2d& w) const;
    Cipher2d conv2d_cipher(const Evaluator& evaluator, const Cipher2d& a, const Cipher2d& w) const;
    // outputs [b][co][H-kh+1][W-kw+1]
    Plain2d encode_outputs_uint64s(const BatchEncoder& encoder, const uint64_t* outputs) const;
    std::vector<uint64_t> decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    // CKKS forms: real-valued tensors through CKKSEncoder's polynomial encoding (decoded values carry scale_x * scale_w)
    Plain2d encode_weights_doubles(const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const;
    Plain2d encode_inputs_doubles(const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const;
    Cipher2d encrypt_weights_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const;
    Cipher2d encrypt_inputs_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const;
    Plain2d encode_outputs_doubles(const CKKSEncoder& encoder, const double* outputs, std::optional<ParmsID> parms_id, double scale) const;
    std::vector<double> decrypt_outputs_doubles(const CKKSEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    // Z_{2^k} tensors through PolynomialEncoderRing2k<T>; for_cipher = the operand that gets encrypted (scaled up) rather than multiplied in (centralized)
    template <typename T> Plain2d encode_weights_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id, bool for_cipher = false) const;
    template <typename T> Plain2d encode_inputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id, bool for_cipher = false) const;
    template <typename T> Cipher2d encrypt_weights_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const;
    template <typename T> Cipher2d encrypt_inputs_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const;
    template <typename T> Plain2d encode_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* outputs, std::optional<ParmsID> parms_id) const;
    template <typename T> std::vector<T> decrypt_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    bool batched_mul = true;          // conv2d.h:49-52: always the case here
    void set_pool(MemoryPoolHandle p) { pool = std::move(p); }
    void serialize_outputs(const Evaluator& evaluator, const Cipher2d& x, std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    Cipher2d deserialize_outputs(const Evaluator& evaluator, std::istream& stream) const;

private:
    void determine_block();
    std::vector<uint64_t> pack_weights(uint64_t t, const uint64_t* weights, size_t& rows, size_t& cols, size_t& len) const;
    std::vector<uint64_t> pack_inputs(uint64_t t, const uint64_t* inputs, size_t& rows, size_t& cols, size_t& len) const;
    template <typename T, typename Check> std::vector<T> pack_weights_of(const T* weights, size_t& rows, size_t& cols, size_t& len, Check&& check) const;
    template <typename T, typename Check> std::vector<T> pack_inputs_of(const T* inputs, size_t& rows, size_t& cols, size_t& len, Check&& check) const;
    template <typename T> Plain2d encode_ring2k_blocks(const PolynomialEncoderRing2k<T>& encoder, const std::vector<T>& packed, size_t rows, size_t cols, size_t len,
                                                       std::optional<ParmsID> parms_id, bool for_cipher) const;
    std::vector<size_t> output_terms() const;
    // calls f(tile, oc_block, coefficient index, output index) for every output element
    template <typename F> void for_each_output(F&& f) const;
};


namespace conv2d_detail { inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; } }

template <typename T, typename Check>
std::vector<T> Conv2dHelper::pack_weights_of(const T* weights, size_t& rows, size_t& cols, size_t& len, Check&& check) const {
    // app/conv2d.cu:110-134: per (output block, input block) the flipped kernels, input channels in reverse order
    const size_t blk = image_height_block * image_width_block;
    rows = conv2d_detail::ceil_div(output_channels, output_channel_block); cols = conv2d_detail::ceil_div(input_channels, input_channel_block);
    len = input_channel_block * output_channel_block * blk;
    std::vector<T> packed(rows * cols * len, 0);
    size_t idx = 0;
    for (size_t loc = 0; loc < output_channels; loc += output_channel_block) {
        const size_t uoc = std::min(loc + output_channel_block, output_channels);
        for (size_t lic = 0; lic < input_channels; lic += input_channel_block, idx++) {
            const size_t uic = std::min(lic + input_channel_block, input_channels);
            T* spread = packed.data() + idx * len;
            for (size_t oc = loc; oc < uoc; oc++)
                for (size_t ic = lic; ic < uic; ic++)
                    for (size_t ki = 0; ki < kernel_height; ki++)
                        for (size_t kj = 0; kj < kernel_width; kj++) {
                            const T v = weights[((oc * input_channels) + ic) * (kernel_height * kernel_width) + (kernel_height - ki - 1) * kernel_width + (kernel_width - kj - 1)];
                            check(v);
                            spread[(oc - loc) * input_channel_block * blk + (input_channel_block - 1 - (ic - lic)) * blk + ki * image_width_block + kj] = v;
                        }
        }
    }
    return packed;
}

template <typename T, typename Check>
std::vector<T> Conv2dHelper::pack_inputs_of(const T* inputs, size_t& rows, size_t& cols, size_t& len, Check&& check) const {
    // app/conv2d.cu:176-222: overlapping tiles (stride hb - (kh-1), wb - (kw-1)), one row of input-channel blocks per tile
    const size_t kh = kernel_height - 1, kw = kernel_width - 1;
    const size_t sh = conv2d_detail::ceil_div(image_height - kh, image_height_block - kh), sw = conv2d_detail::ceil_div(image_width - kw, image_width_block - kw);
    const size_t image_size = image_height * image_width, blk = image_height_block * image_width_block;
    rows = conv2d_detail::ceil_div(batch_size, batch_block) * sh * sw; cols = conv2d_detail::ceil_div(input_channels, input_channel_block); len = slot_count;
    std::vector<T> packed(rows * cols * len, 0);
    size_t idx = 0;
    for (size_t lb = 0; lb < batch_size; lb += batch_block) {
        const size_t ub = std::min(lb + batch_block, batch_size);
        for (size_t ih = 0; ih < sh; ih++)
            for (size_t iw = 0; iw < sw; iw++) {
                const size_t si = ih * (image_height_block - kh), sj = iw * (image_width_block - kw);
                const size_t ui = std::min(si + image_height_block, image_height), uj = std::min(sj + image_width_block, image_width);
                for (size_t lci = 0; lci < input_channels; lci += input_channel_block, idx++) {
                    const size_t uci = std::min(lci + input_channel_block, input_channels);
                    T* vec = packed.data() + idx * len;
                    for (size_t b = 0; b < ub - lb; b++)
                        for (size_t tci = 0; tci < uci - lci; tci++)
                            for (size_t ti = si; ti < ui; ti++)
                                for (size_t tj = sj; tj < uj; tj++) {
                                    const T v = inputs[(lb + b) * input_channels * image_size + (lci + tci) * image_size + ti * image_width + tj];
                                    check(v);
                                    vec[b * input_channel_block * output_channel_block * blk + tci * blk + (ti - si) * image_width_block + (tj - sj)] = v;
                                }
                }
            }
    }
    return packed;
}

template <typename F>
void Conv2dHelper::for_each_output(F&& f) const {
    // app/conv2d.cu:259-292, :306-345: tile (ob, si, sj), output block lc, then the (b, c, i, j) of the tile that exist in the image
    const size_t interval = image_width_block * image_height_block;
    const size_t yh = image_height_block - kernel_height + 1, yw = image_width_block - kernel_width + 1;
    const size_t oyh = image_height - kernel_height + 1, oyw = image_width - kernel_width + 1;
    const size_t kh = kernel_height - 1, kw = kernel_width - 1;
    const size_t sh = conv2d_detail::ceil_div(image_height - kh, image_height_block - kh), sw = conv2d_detail::ceil_div(image_width - kw, image_width_block - kw);
    const size_t tiles = get_total_batch_size();
    for (size_t eb = 0; eb < tiles; eb++) {
        const size_t ob = eb / (sh * sw), si = (eb % (sh * sw)) / sw, sj = eb % sw;
        const size_t lb = ob * batch_block, ub = std::min(lb + batch_block, batch_size);
        for (size_t lc = 0; lc < output_channels; lc += output_channel_block) {
            const size_t uc = std::min(lc + output_channel_block, output_channels);
            for (size_t b = lb; b < ub; b++)
                for (size_t c = lc; c < uc; c++)
                    for (size_t i = 0; i < yh; i++)
                        for (size_t j = 0; j < yw; j++) {
                            if (si * yh + i >= oyh || sj * yw + j >= oyw) continue;
                            const size_t coefficient = ((b - lb) * input_channel_block * output_channel_block + (c - lc) * input_channel_block + input_channel_block - 1) * interval +
                                                       (image_height_block - yh + i) * image_width_block + (image_width_block - yw + j);
                            f(eb, lc / output_channel_block, coefficient, b * output_channels * oyh * oyw + c * oyh * oyw + (si * yh + i) * oyw + (sj * yw + j));
                        }
        }
    }
}

// ---- ring-2^k forms (app/conv2d with the Ring2k encoder adapter): the encrypted operand is scaled up, the plaintext operand centralized ----
template <typename T>
Plain2d Conv2dHelper::encode_ring2k_blocks(const PolynomialEncoderRing2k<T>& encoder, const std::vector<T>& packed, size_t rows, size_t cols, size_t len, std::optional<ParmsID> parms_id,
                                           bool for_cipher) const {
    Evaluator evaluator(encoder.context());
    Plain2d out;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t k = 0; k < cols; k++) {
            const auto begin = packed.begin() + static_cast<std::ptrdiff_t>((r * cols + k) * len);
            const std::vector<T> block(begin, begin + static_cast<std::ptrdiff_t>(len));
            Plaintext p = for_cipher ? encoder.scale_up_new(block, parms_id, pool) : encoder.centralize_new(block, parms_id, pool);
            evaluator.transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(std::move(p));
        }
    }
    return out;
}
template <typename T>
Plain2d Conv2dHelper::encode_weights_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id, bool for_cipher) const {
    size_t rows, cols, len;
    const std::vector<T> packed = pack_weights_of<T>(weights, rows, cols, len, [](T) {});
    return encode_ring2k_blocks(encoder, packed, rows, cols, len, parms_id, for_cipher);
}
template <typename T>
Plain2d Conv2dHelper::encode_inputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id, bool for_cipher) const {
    size_t rows, cols, len;
    const std::vector<T> packed = pack_inputs_of<T>(inputs, rows, cols, len, [](T) {});
    return encode_ring2k_blocks(encoder, packed, rows, cols, len, parms_id, for_cipher);
}
template <typename T>
Cipher2d Conv2dHelper::encrypt_weights_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const {
    const Plain2d plain = encode_weights_ring2k(encoder, weights, parms_id, true);
    Cipher2d out;
    for (const auto& prow : plain.data()) { std::vector<Ciphertext>& row = out.new_row(); for (const Plaintext& p : prow) row.push_back(encryptor.encrypt_symmetric_new(p, true, pool)); }
    return out;
}
template <typename T>
Cipher2d Conv2dHelper::encrypt_inputs_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const {
    const Plain2d plain = encode_inputs_ring2k(encoder, inputs, parms_id, true);
    Cipher2d out;
    for (const auto& prow : plain.data()) { std::vector<Ciphertext>& row = out.new_row(); for (const Plaintext& p : prow) row.push_back(encryptor.encrypt_symmetric_new(p, true, pool)); }
    return out;
}
template <typename T>
Plain2d Conv2dHelper::encode_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* outputs, std::optional<ParmsID> parms_id) const {
    const size_t tiles = get_total_batch_size(), groups = conv2d_detail::ceil_div(output_channels, output_channel_block);
    std::vector<std::vector<T>> buffers(tiles * groups, std::vector<T>(slot_count, 0));
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { buffers[tile * groups + group][coefficient] = outputs[index]; });
    Plain2d out;
    for (size_t tile = 0; tile < tiles; tile++) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t g = 0; g < groups; g++) row.push_back(encoder.scale_up_new(buffers[tile * groups + g], parms_id, pool));
    }
    return out;
}
template <typename T>
std::vector<T> Conv2dHelper::decrypt_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    const size_t tiles = get_total_batch_size(), groups = conv2d_detail::ceil_div(output_channels, output_channel_block);
    std::vector<std::vector<T>> coeffs;
    for (const auto& r : outputs.data())
        for (const Ciphertext& ct : r) coeffs.push_back(encoder.scale_down_new(decryptor.bfv_decrypt_without_scaling_down_new(ct, pool), pool));
    if (coeffs.size() != tiles * groups) throw std::invalid_argument("[Conv2dHelper::decrypt_outputs] Output ciphertext count incorrect");
    const size_t oyh = image_height - kernel_height + 1, oyw = image_width - kernel_width + 1;
    std::vector<T> out(batch_size * output_channels * oyh * oyw, 0);
    for_each_output([&](size_t tile, size_t group, size_t coefficient, size_t index) { out[index] = coeffs[tile * groups + group][coefficient]; });
    return out;
}

inline std::ostream& operator<<(std::ostream& os, const Conv2dHelper& h) {               // conv2d.cu:5-18
    return os << "Conv2dHelper(batch_size=" << h.batch_size << ", input_channels=" << h.input_channels << ", output_channels=" << h.output_channels << ", image_height=" << h.image_height
              << ", image_width=" << h.image_width << ", kernel_height=" << h.kernel_height << ", kernel_width=" << h.kernel_width << ", slot_count=" << h.slot_count
              << ", objective=" << h.objective << ")";
}

}}  // namespace troy::linear

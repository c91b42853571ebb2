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
    void serialize_outputs(const Evaluator& evaluator, const Cipher2d& x, std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    Cipher2d deserialize_outputs(const Evaluator& evaluator, std::istream& stream) const;

private:
    void determine_block();
    std::vector<uint64_t> pack_weights(uint64_t t, const uint64_t* weights, size_t& rows, size_t& cols, size_t& len) const;
    std::vector<uint64_t> pack_inputs(uint64_t t, const uint64_t* inputs, size_t& rows, size_t& cols, size_t& len) const;
    std::vector<size_t> output_terms() const;
    // calls f(tile, oc_block, coefficient index, output index) for every output element
    template <typename F> void for_each_output(F&& f) const;
};

}}  // namespace troy::linear

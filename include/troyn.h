/*
 * troyn.h -- C-ABI of the MI355X-native RNS-RLWE hot path (libtroyn.so).
 *
 * This is the drop-in boundary for the data-parallel hot path of lightbulb128/troy-nova:
 * negacyclic NTT/INTT, RNS dyadic multiply/add, key switching (relinearize), modulus
 * switching / CKKS rescale, and the BEHZ BFV multiply.  The reference has no FFI layer for
 * this path (its device code is reached by `if (on_device()) kernel<<<>>>` branches inside
 * C++ functions), so every entry point below names the reference C++
 * function (file:line under the reference's src/) whose *device branch* it replaces.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes only.  All polynomial data are DEVICE pointers to uint64_t in
 *    the reference layout data[(p*L + l)*N + i] (ciphertext.h:211-247), canonical residues.
 *  - every op takes `batch` independent items laid out contiguously ([batch][...]) unless a
 *    stride is given explicitly; batch = 1 reproduces the reference's single-object calls,
 *    batch > 1 replaces its `*_batched` pointer-table variants (box_batch.h:231-257).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *    work; they never synchronise and never allocate device memory.  Temporaries come from a
 *    caller-supplied workspace (size from the matching *_workspace_bytes query), which is how
 *    the reference's MemoryPool plugs in.
 *  - return value: 0 = success; > 0 = hipError_t from the runtime; < 0 = TROYN_E_* below.
 *    troyn_last_error() returns a thread-local message (the C++ mirror turns negative codes
 *    into std::invalid_argument and positive codes into std::runtime_error, matching
 *    kernel_provider.h:11-16 and the reference's "[Class::method] ..." messages).
 */
#ifndef TROYN_H
#define TROYN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TROYN_VERSION 1

enum {
    TROYN_OK = 0,
    TROYN_E_INVALID = -1,     /* bad argument (sizes, null pointers, unsupported N) */
    TROYN_E_MODULUS = -2,     /* modulus not usable (not < 2^61, no 2N-th root, not coprime) */
    TROYN_E_WORKSPACE = -3,   /* workspace too small */
    TROYN_E_UNSUPPORTED = -4  /* scheme / shape not implemented */
};

typedef void* troyn_stream_t;
typedef struct troyn_plan troyn_plan;
typedef struct troyn_behz troyn_behz;

const char* troyn_last_error(void);
int troyn_version(void);

/* Measurement hook (the reference times its kernels with bench::Timer around whole calls, test/bench/he_operations.cu:57-104;
 * a fused call here contains several launches, so the library can bracket ONE named launch with hipEvents on the launch stream).
 *   troyn_kernel_timer_enable(region, on)   start / stop recording an event pair around every launch of that region
 *   troyn_kernel_timer_read(region, &ms, &n) wait for the recorded events, return their summed duration and count, and clear
 * Region TROYN_TIMER_KS_INNER_PRODUCT = the fused key-switch inner product (kernel_set_accumulate + ntt +
 * kernel_accumulate_products of fgk/switch_key.cu, one launch here);
 * TROYN_TIMER_BFV_TENSOR = the transform + tensor-product launches of bfv_multiply (ntt_inplace_ps x2 + kernel_dyadic_convolute +
 * intt_inplace_ps, evaluator.cu:56-93; one launch per base and arithmetic class here);
 * TROYN_TIMER_PLAIN_MAC = the ct x pt multiply-accumulate launch (multiply_plain_ntt_accumulate, evaluator_multiply_plain.cu:356-385);
 * TROYN_TIMER_BEHZ_FLOOR = fast_floor_fast_b_conv_sk (fgk/rns_tool.cu:147-343). */
enum { TROYN_TIMER_KS_INNER_PRODUCT = 0, TROYN_TIMER_BFV_TENSOR = 1, TROYN_TIMER_PLAIN_MAC = 2, TROYN_TIMER_BEHZ_FLOOR = 3, TROYN_TIMER_REGIONS = 4 };
int troyn_kernel_timer_enable(int region, int on);
int troyn_kernel_timer_read(int region, double* total_ms, uint64_t* launches);

/* Host-only helpers (no GPU touched) so that a host can build the same parameter sets as the
 * reference: CoeffModulus::create (coeff_modulus.cu:65-108: for every bit size the k largest primes
 * = 1 mod 2N, handed out smallest-first) and utils::get_primes (utils/number_theory.cu:22-39). */
int troyn_coeff_modulus_create(size_t poly_modulus_degree, const size_t* bit_sizes, size_t n, uint64_t* out);
int troyn_get_primes(uint64_t factor, size_t bit_size, size_t count, uint64_t* out);

/* ---------------------------------------------------------------------------------------
 * Plan = device-resident mirror of one modulus chain: for each of the `n_moduli` primes
 * (key level order, special prime last) the Modulus constants (modulus.h:8-124), the
 * NTTTables (utils/ntt.h:12-87, built exactly as utils/ntt.cu:14-76 does: minimal primitive
 * 2N-th root, bit-reversed psi powers, scrambled inverse powers, N^-1) and, for every level
 * L <= n_moduli, q_{L-1}^-1 mod q_i (RNSTool::inv_q_last_mod_q, utils/rns_tool.cu:225-236).
 * Replaces ContextData::to_device_inplace (context_data.cu:34-69).
 * `roots` may be NULL (roots are then searched as number_theory.cu:68-87 does) or hold the
 * reference's NTTTables::root() per modulus.
 * ------------------------------------------------------------------------------------- */
int troyn_plan_create(troyn_plan** plan, int device, uint32_t log_n, uint32_t n_moduli,
                      const uint64_t* moduli, const uint64_t* roots);
int troyn_plan_destroy(troyn_plan* plan);
/* A/B switches (DESIGN.md section 4 "A/B switches"): read from the environment variables of the same names ONCE, when the plan is created;
 * this sets one of them on an existing plan ("TROYN_KS_ORDER", "row"; value NULL or "" = the default).  The library never reads the
 * environment on a call path.  Not to be called while other threads use the plan.  Replaces nothing in the reference (development hook). */
int troyn_plan_set_option(troyn_plan* plan, const char* name, const char* value);
uint32_t troyn_plan_log_n(const troyn_plan* plan);
uint32_t troyn_plan_n_moduli(const troyn_plan* plan);
/* host copies of table contents, for known-answer checks: out[2*i] = operand, out[2*i+1] = quotient */
int troyn_plan_get_root(const troyn_plan* plan, uint32_t modulus_index, uint64_t* root);
int troyn_plan_get_root_powers(const troyn_plan* plan, uint32_t modulus_index, int inverse, uint64_t* out);

/* NTTTableIndexer modes (utils/ntt.h:89-130) */
enum { TROYN_IDX_COMPONENTWISE = 0, TROYN_IDX_KS_SET_PRODUCTS = 1, TROYN_IDX_KS_SKIP_FINALS = 2 };

/* ---------------------------------------------------------------------------------------
 * troyn_ntt: negacyclic NTT (inverse = 0; natural -> bit-reversed, canonical output) or INTT
 * (inverse = 1; includes the N^-1 scaling) of batch*pcount*ncomp limb-polynomials.
 * Replaces the device branches of fgk::ntt_grouped::ntt / intt (fgk/ntt_grouped.cu:258-295,
 * :597-640) and their _batched forms, i.e. utils::ntt_ps / intt_ps / *_inplace_* /
 * *_b* (utils/ntt.h:164-391).  in == out is allowed.  Component j of polynomial k uses table
 * `table_start + get(k, j)` where get() is NTTTableIndexer::get over a slice of
 * `table_count` tables (utils/ntt.h:105-124).
 * ------------------------------------------------------------------------------------- */
int troyn_ntt(const troyn_plan* plan, int inverse, const uint64_t* in, uint64_t* out,
              size_t batch, size_t pcount, size_t ncomp,
              uint32_t table_start, uint32_t table_count, int indexer_mode, uint32_t decomp_size,
              troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Element-wise RNS polynomial ops over count*nmod limb-polynomials ([count][nmod][N]); limb l
 * uses modulus mod_start + l.  Replace the device branches of utils::add_ps / sub_ps /
 * negate_ps / multiply_scalar_ps / dyadic_product_ps / modulo_ps / multiply_uint64operand_ps
 * (utils/poly_small_mod.cu:243-304, :306-365, :182-241, :653-714, :816-900, :119-180, :752-814).
 * troyn_modulo: Modulus::reduce (Barrett-64, modulus.h:22-42) of ANY 64-bit word.
 * troyn_multiply_uint64operand: `operands` = nmod device-resident (operand, quotient) pairs, one
 * MultiplyUint64Operand (utils/uint_small_mod.h:92-122) per limb of the slice; the input may be any
 * 64-bit word, the result is canonical (multiply_uint64operand_mod, :130-139).
 * ------------------------------------------------------------------------------------- */
int troyn_add(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a, const uint64_t* b,
              uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_sub(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a, const uint64_t* b,
              uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_negate(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a,
                 uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_multiply_scalar(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a, uint64_t scalar,
                          uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_dyadic_product(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a, const uint64_t* b,
                         uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_modulo(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a,
                 uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_multiply_uint64operand(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* a,
                                 const uint64_t* operands, uint64_t* out, size_t count, troyn_stream_t stream);

/* fgk::dyadic_convolute::dyadic_convolute (fgk/dyadic_convolute.cu:43-90): a[pa][nmod][N] x
 * b[pb][nmod][N] -> out[pa+pb-1][nmod][N], per batch item.  CKKS/BGV multiply is exactly this
 * (evaluator.cu:118-173). */
int troyn_dyadic_convolute(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod,
                           const uint64_t* a, size_t pa, const uint64_t* b, size_t pb, uint64_t* out,
                           size_t batch, troyn_stream_t stream);
/* fgk::dyadic_convolute::dyadic_square (fgk/dyadic_convolute.cu:116-150) */
int troyn_dyadic_square(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod,
                        const uint64_t* a, uint64_t* out, size_t batch, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Key switching core: Evaluator::switch_key_internal(_batched)
 * (evaluator_keyswitching_core.cu:757-1052, :1055-1267), device branch, BFV/CKKS.
 *   L            decomposition size = limbs of the data level (L <= n_moduli - 1)
 *   target       [batch][L][N]   polynomial to switch (NTT form iff is_ntt_form)
 *   keys         host array of L DEVICE pointers, keys[j] -> u64[2][K][N] in NTT form
 *                (KSwitchKeys::get_data_ptrs, kswitch_keys.h:34-54); shared by the batch
 *   destination  [batch][2][L][N], written or accumulated per `assign_method`
 *                (SwitchKeyDestinationAssignMethod, evaluator.h)
 *   workspace    query it with the batch the call will use: small launches (a single ciphertext) take the digit-parallel
 *                inner product (one workgroup per digit on the caller's own keys, no key preparation pass), whose slots are
 *                part of the workspace; the result words do not depend on which form runs
 * ------------------------------------------------------------------------------------- */
enum { TROYN_ASSIGN_ADD_INPLACE = 0, TROYN_ASSIGN_OVERWRITE = 1, TROYN_ASSIGN_OVERWRITE_EXCEPT_FIRST = 2 };
size_t troyn_switch_key_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch);
int troyn_switch_key(const troyn_plan* plan, uint32_t L, int is_ckks, int is_ntt_form,
                     const uint64_t* target, const uint64_t* const* keys, int assign_method,
                     uint64_t* destination, void* workspace, size_t workspace_bytes,
                     size_t batch, troyn_stream_t stream);

/* Evaluator::relinearize_internal for a 3-polynomial ciphertext (evaluator_keyswitching.cu:
 * 119-144): ct[batch][3][L][N] -> out[batch][2][L][N] = switch_key(ct[2], Overwrite) + ct[0..2). */
size_t troyn_relinearize_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch);
int troyn_relinearize(const troyn_plan* plan, uint32_t L, int is_ckks, int is_ntt_form,
                      const uint64_t* ct3, const uint64_t* const* keys, uint64_t* out2,
                      void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused CKKS chain: Evaluator::multiply (evaluator.cu:118-145) -> Evaluator::relinearize (evaluator_keyswitching.cu:119-144)
 * -> Evaluator::rescale_to_next (evaluator_modswitch.cu, RNSTool::divide_and_round_q_last_ntt utils/rns_tool.cu:499-694) of
 * `batch` ciphertext pairs in one call.  Results are bit-identical to troyn_dyadic_convolute + troyn_relinearize +
 * troyn_divide_and_round_q_last_ntt; on whole-limb FP64 rings (N = 8192 / 16384, moduli < 2^50) the tensor product is formed
 * inside the loaders of the transforms that consume it and the rounding steps of the key switch and of the rescale share one
 * forward transform per output limb; other shapes run the three calls.
 *   a, b  [batch][2][L][N] NTT form;  keys as troyn_switch_key;  out [batch][2][L-1][N] NTT form
 * ------------------------------------------------------------------------------------- */
size_t troyn_ckks_multiply_relinearize_rescale_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch);
int troyn_ckks_multiply_relinearize_rescale(const troyn_plan* plan, uint32_t L, const uint64_t* a, const uint64_t* b,
                                            const uint64_t* const* keys, uint64_t* out, void* workspace, size_t workspace_bytes,
                                            size_t batch, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Modulus switching.
 * troyn_divide_and_round_q_last      RNSTool::divide_and_round_q_last (utils/rns_tool.cu:374-466),
 *                                    BFV mod_switch_to_next, coefficient form.
 * troyn_divide_and_round_q_last_ntt  RNSTool::divide_and_round_q_last_ntt (utils/rns_tool.cu:499-694),
 *                                    CKKS rescale_to_next, NTT form.
 * troyn_mod_switch_drop              kernel_mod_switch_drop_to (evaluator_modswitch.cu:164-171).
 * in [batch][pcount][L][N] -> out [batch][pcount][L-1][N] (drop: [L_out]).
 * ------------------------------------------------------------------------------------- */
int troyn_divide_and_round_q_last(const troyn_plan* plan, uint32_t L, const uint64_t* in, size_t pcount,
                                  uint64_t* out, size_t batch, troyn_stream_t stream);
size_t troyn_divide_and_round_q_last_ntt_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t pcount, size_t batch);
int troyn_divide_and_round_q_last_ntt(const troyn_plan* plan, uint32_t L, const uint64_t* in, size_t pcount,
                                      uint64_t* out, void* workspace, size_t workspace_bytes,
                                      size_t batch, troyn_stream_t stream);
int troyn_mod_switch_drop(const troyn_plan* plan, uint32_t L_in, uint32_t L_out, const uint64_t* in, size_t pcount,
                          uint64_t* out, size_t batch, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Callers either side of the evaluator path, kept on the device:
 *
 * Context PRNG = AES-128 in counter mode (utils/random_generator.cu:37-58,:112-117): key = seed[2]
 * (low, high), block i = AES(counter + i).  The samplers write `nmod` limbs (the first nmod plan
 * moduli) of one polynomial and report how many blocks they consumed so the caller can advance its
 * counter exactly like RandomGenerator does:
 *   troyn_sample_ternary            RandomGenerator::sample_poly_ternary            (:318-336)  ceil(N/16) blocks
 *   troyn_sample_centered_binomial  RandomGenerator::sample_poly_centered_binomial  (:421-440)  ceil(N/2)  blocks
 *   troyn_sample_uniform            RandomGenerator::sample_poly_uniform            (:475-481)  ceil(nmod*N/2) blocks
 *   troyn_prng_block                host_generate_uint128 (:112-117), host only: sample_uint64() = out[0]
 * ------------------------------------------------------------------------------------- */
int troyn_prng_block(const uint64_t seed[2], uint64_t counter, uint64_t out[2]);
int troyn_sample_ternary(const troyn_plan* plan, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                         uint64_t* blocks_used, troyn_stream_t stream);
int troyn_sample_centered_binomial(const troyn_plan* plan, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                                   uint64_t* blocks_used, troyn_stream_t stream);
int troyn_sample_uniform(const troyn_plan* plan, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                         uint64_t* blocks_used, troyn_stream_t stream);
/* batched forms (one launch for `count` polynomials, out[count][nmod][N]):
 *   _strided: polynomial i continues the SAME generator at counter + i*counter_stride -- the positions `count`
 *             sequential encryptions would have used (each consumes ceil(N/2) blocks here plus whatever else it draws);
 *   _multi:   polynomial i comes from its own generator seeds[2i], seeds[2i+1] at counter 0 (the per-ciphertext
 *             c1 generators of rlwe::symmetric, utils/rlwe.cu:262-266). */
int troyn_sample_centered_binomial_strided(const troyn_plan* plan, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t counter_stride,
                                           uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_sample_uniform_multi(const troyn_plan* plan, uint32_t nmod, const uint64_t* seeds, uint64_t* out, size_t count, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Ciphertext x plaintext (the BASELINE config 5 matmul path):
 *   troyn_plain_centralize        scaling_variant::centralize (utils/scaling_variant.cu:326-357, fast-plain-lift case):
 *                                 plain[batch][count] mod t -> dest[batch][L][N], ready for the forward NTT
 *                                 (Evaluator::transform_plain_to_ntt, evaluator_transform_ntt.cu:35-70)
 *   troyn_dyadic_broadcast_product fgk::dyadic_convolute::dyadic_broadcast_product_ps (fgk/dyadic_convolute.cu:173-195):
 *                                 out[b][p][l] = ct[b][p][l] (.) pt[b][l]; pt_bstride = 0 shares one plaintext
 *   troyn_multiply_plain_accumulate Evaluator::multiply_plain_ntt_accumulate (evaluator_multiply_plain.cu:258-307):
 *                                 dst[k] (+)= ct[k] (.) pt[k] over `count` triples of device pointers (host arrays);
 *                                 equal dst pointers accumulate; set_zero != 0 overwrites instead of adding to dst.
 * ------------------------------------------------------------------------------------- */
int troyn_plain_centralize(const troyn_plan* plan, uint32_t L, uint64_t plain_modulus, const uint64_t* plain, size_t plain_coeff_count,
                           size_t plain_bstride, uint64_t* dest, size_t batch, troyn_stream_t stream);
/* Evaluator::transform_plain_to_ntt (evaluator_transform_ntt.cu:35-70) as ONE launch: scaling_variant::centralize (utils/scaling_variant.cu:326-357) in the
 * loader of the forward transform -- dest[batch][L][N] = NTT_j(centralize_j(plain[b])), the words troyn_plain_centralize + troyn_ntt give; the centred
 * coefficient-form polynomial is never written and the zeros beyond plain_coeff_count are never read.  (Falls back to the two launches where the fused loader
 * does not apply: N < 1024, or t not below every modulus.) */
int troyn_plain_centralize_ntt(const troyn_plan* plan, uint32_t L, uint64_t plain_modulus, const uint64_t* plain, size_t plain_coeff_count,
                               size_t plain_bstride, uint64_t* dest, size_t batch, troyn_stream_t stream);
int troyn_dyadic_broadcast_product(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* ct, size_t pcount,
                                   const uint64_t* pt, size_t pt_bstride, uint64_t* out, size_t batch, troyn_stream_t stream);
/* Galois automorphism X -> X^g of `count` polynomials of nmod limbs each:
 * GaloisTool::apply_ps (coefficient form, utils/galois.cu:168-206) / apply_ntt_ps (NTT form, :24-41,:250-343).
 * Out of place (in != out).  Evaluator::apply_galois = this permutation of (c0, c1) followed by troyn_switch_key on
 * the permuted c1 with TROYN_ASSIGN_OVERWRITE_EXCEPT_FIRST (evaluator_keyswitching.cu:147-179). */
int troyn_apply_galois(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, int is_ntt_form, uint64_t galois_element,
                       const uint64_t* in, uint64_t* out, size_t count, troyn_stream_t stream);
/* GaloisTool::apply (utils/galois.cu:43-66) on `count` coefficient-form polynomials [N] modulo ONE explicit modulus -- the plain
 * modulus t of a BFV / BGV plaintext (Evaluator::apply_galois_plain, evaluator_keyswitching.cu:235-261). */
int troyn_apply_galois_plain(const troyn_plan* plan, uint64_t modulus, uint64_t galois_element, const uint64_t* in, uint64_t* out, size_t count,
                             troyn_stream_t stream);
/* ---------------------------------------------------------------------------------------
 * BGV.  BGV ciphertexts live in NTT form and reuse troyn_ntt, troyn_dyadic_convolute (bgv_multiply,
 * evaluator.cu:150-173), troyn_add/sub/negate/multiply_scalar, troyn_apply_galois and troyn_plain_centralize unchanged; the
 * entries below are the steps where BGV differs.  troyn_bgv holds the constants of ONE level's RNSTool that only BGV reads
 * (utils/rns_tool.cu:205-232): the converter q -> {t}, q_last^-1 mod t.
 *   troyn_bgv_mod_t_and_divide_q_last_ntt  RNSTool::mod_t_and_divide_q_last_ntt (utils/rns_tool.cu:1540-1772):
 *                                   in [batch][pcount][L][N] NTT -> out [batch][pcount][L-1][N] NTT; the caller multiplies
 *                                   the correction factor by troyn_bgv_inv_q_last_mod_t (evaluator_modswitch.cu:70-72)
 *   troyn_bgv_decrypt_mod_t         scaling_variant::decentralize (utils/scaling_variant.cu:415-431) = BaseConverter::
 *                                   exact_convey_array (utils/rns_base.cu:445-598, double-precision vote summed in limb
 *                                   order) then * correction_factor^-1 mod t: phase [batch][L][N] coefficient form -> [batch][N]
 *   troyn_bgv_multiply_scalar_mod_t utils::multiply_scalar on mod-t plaintext words (evaluator_translate_plain.cu:80-82)
 *   troyn_bgv_switch_key / _relinearize  switch_key_internal / relinearize_internal with the ski_util5 tail
 *                                   (evaluator_keyswitching_core.cu:436-538, :998-1030); `key_level` must be created with
 *                                   L = the plan's modulus count (its last prime is the special prime).  Shapes and
 *                                   workspace as troyn_switch_key / troyn_relinearize, NTT-form operands.
 * ------------------------------------------------------------------------------------- */
typedef struct troyn_bgv troyn_bgv;
int troyn_bgv_create(troyn_bgv** out, const troyn_plan* plan, uint32_t L, uint64_t plain_modulus);
int troyn_bgv_destroy(troyn_bgv* bgv);
uint64_t troyn_bgv_inv_q_last_mod_t(const troyn_bgv* bgv);
size_t troyn_bgv_mod_switch_workspace_bytes(const troyn_bgv* bgv, size_t pcount, size_t batch);
int troyn_bgv_mod_t_and_divide_q_last_ntt(const troyn_bgv* bgv, const uint64_t* in, size_t pcount, uint64_t* out, void* workspace, size_t workspace_bytes,
                                          size_t batch, troyn_stream_t stream);
int troyn_bgv_decrypt_mod_t(const troyn_bgv* bgv, const uint64_t* phase, uint64_t correction_factor, uint64_t* dest, size_t batch, troyn_stream_t stream);
int troyn_bgv_multiply_scalar_mod_t(const troyn_bgv* bgv, const uint64_t* in, uint64_t scalar, uint64_t* out, size_t count, troyn_stream_t stream);
int troyn_bgv_switch_key(const troyn_bgv* key_level, uint32_t L, const uint64_t* target, const uint64_t* const* keys, int assign_method,
                         uint64_t* destination, void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream);
int troyn_bgv_relinearize(const troyn_bgv* key_level, uint32_t L, const uint64_t* ct3, const uint64_t* const* keys, uint64_t* out2, void* workspace,
                          size_t workspace_bytes, size_t batch, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Ring-2^k polynomial encoder (src/app/bfv_ring2k.{h,cu}: PolynomialEncoderRNSHelper<T>, T = uint32_t / uint64_t / uint128_t):
 * plaintext modulus t = 2^k outside the context's own plain modulus.  One handle per level (first L primes), k and element width.
 *   troyn_ring2k_scale_up     scale_up (:299-360): src elements [count] -> out u64[L][N] = round(Q/t * m), zero beyond `count`
 *   troyn_ring2k_centralize   centralize (:488-560): centred lift of m
 *   troyn_ring2k_scale_down   scale_down (:620-735): in u64[L][N] (phase, coefficient form) -> dst elements [N]
 *   troyn_ring2k_decentralize decentralize (:752-911): in u64[L][N] (x mod Q, coefficient form) -> dst elements [N] = x mod 2^k, times correction^-1 mod 2^k
 *                             (correction_lo / _hi: the odd correction factor as a 128-bit value; 1, 0 for none)
 * ------------------------------------------------------------------------------------- */
typedef struct troyn_ring2k troyn_ring2k;
int troyn_ring2k_create(troyn_ring2k** out, const troyn_plan* plan, uint32_t L, uint32_t t_bit_length, uint32_t element_bytes);
int troyn_ring2k_destroy(troyn_ring2k* h);
uint64_t troyn_ring2k_gamma(const troyn_ring2k* h);
int troyn_ring2k_scale_up(const troyn_ring2k* h, const void* src, size_t count, uint64_t* out, troyn_stream_t stream);
int troyn_ring2k_centralize(const troyn_ring2k* h, const void* src, size_t count, uint64_t* out, troyn_stream_t stream);
int troyn_ring2k_scale_down(const troyn_ring2k* h, const uint64_t* in, void* dst, troyn_stream_t stream);
int troyn_ring2k_decentralize(const troyn_ring2k* h, const uint64_t* in, void* dst, uint64_t correction_lo, uint64_t correction_hi, troyn_stream_t stream);

/* RLWE / LWE packing (evaluator_lwes.cu):
 *   troyn_negacyclic_shift     utils::negacyclic_shift_ps (utils/poly_small_mod.cu:927-968): multiply `count` RNS polynomials by
 *                              X^shift, any shift (taken modulo 2N as the reference does); out of place
 *   troyn_multiply_inv_degree  utils::ntt_multiply_inv_degree (utils/ntt.cu:93-134): x * N^-1 * scalar mod q_l
 *                              (Evaluator::divide_by_poly_modulus_degree_inplace, evaluator_lwes.cu:142-151)
 *   troyn_pack_prepare         first step of Evaluator::pack_rlwe_ciphertexts_new(_batched) (evaluator_lwes.cu:361-381, :599-640):
 *                              out[slot] = X^shift * src[slot] * N^-1 * mul, zero where src[slot] == NULL; src = host array of
 *                              `slots` device pointers to coefficient-form ciphertexts u64[pcount][L][N]
 *   troyn_pack_layer           one layer of the packing tree (:441-477, :654-697) on adjacent pairs in[2k] (even), in[2k+1] (odd)
 *                              of two-polynomial ciphertexts:  temp = X^shift*odd, out[k] = even + temp + perm_g(even - temp) with
 *                              the permuted c1 diverted to target[k]; follow with troyn_switch_key(target -> out, ADD_INPLACE,
 *                              batch = pairs, Galois key of g) to finish apply_galois.  in u64[2*pairs][2][L][N],
 *                              out u64[pairs][2][L][N], target u64[pairs][L][N]
 *   troyn_extract_lwe          Evaluator::extract_lwe_new (evaluator_lwes.cu:52-97) for `count` (ciphertext, term) pairs:
 *                              c0 u64[count][L], c1 u64[count][L][N] */
int troyn_negacyclic_shift(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* in, uint64_t* out, size_t shift,
                           size_t count, troyn_stream_t stream);
int troyn_multiply_inv_degree(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, const uint64_t* in, uint64_t* out, uint64_t scalar,
                              size_t count, troyn_stream_t stream);
size_t troyn_pack_prepare_workspace_bytes(size_t slots);
int troyn_pack_prepare(const troyn_plan* plan, uint32_t L, size_t pcount, const uint64_t* const* src, size_t slots, uint64_t mul, size_t shift,
                       uint64_t* out, void* workspace, size_t workspace_bytes, troyn_stream_t stream);
int troyn_pack_layer(const troyn_plan* plan, uint32_t L, uint64_t galois_element, size_t shift, const uint64_t* in, uint64_t* out,
                     uint64_t* target, size_t pairs, troyn_stream_t stream);
size_t troyn_extract_lwe_workspace_bytes(size_t count);
int troyn_extract_lwe(const troyn_plan* plan, uint32_t L, const uint64_t* const* ct, const size_t* terms, uint64_t* c0, uint64_t* c1, size_t count,
                      void* workspace, size_t workspace_bytes, troyn_stream_t stream);
/* Staging for the reference's x_batched(vector<const T*>, ...) forms (batch_utils.h construct_batch + per-item loops inside
 * the kernels): `count` scattered device buffers of `words` words (16-byte aligned) -> one contiguous block, one launch. */
size_t troyn_gather_workspace_bytes(size_t count);
int troyn_gather(const uint64_t* const* src, size_t count, size_t words, uint64_t* out, void* workspace, size_t workspace_bytes, troyn_stream_t stream);
/* The inverse (results of one batched call back to `count` separate buffers; workspace as for troyn_gather).  Up to 64 buffers the
 * pointers of either call travel in the kernel arguments: no upload, no host wait. */
int troyn_scatter(const uint64_t* in, uint64_t* const* dst, size_t count, size_t words, void* workspace, size_t workspace_bytes, troyn_stream_t stream);
size_t troyn_multiply_plain_accumulate_workspace_bytes(size_t count);
int troyn_multiply_plain_accumulate(const troyn_plan* plan, uint32_t mod_start, uint32_t nmod, size_t pcount,
                                    const uint64_t* const* ct, const uint64_t* const* pt, uint64_t* const* dst, size_t count,
                                    int set_zero, void* workspace, size_t workspace_bytes, troyn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * BEHZ BFV multiply: Evaluator::bfv_multiply (evaluator.cu:29-116) with the RNSTool of the level
 * holding the first L plan moduli and plain modulus t (RNSTool ctor utils/rns_tool.cu:29-275;
 * fused device kernels fgk/rns_tool.cu:7-100, :147-286).  a[batch][pa][L][N] x b[batch][pb][L][N]
 * (coefficient form) -> out[batch][pa+pb-1][L][N].
 * ------------------------------------------------------------------------------------- */
int troyn_behz_create(troyn_behz** behz, const troyn_plan* plan, uint32_t L, uint64_t plain_modulus);
int troyn_behz_destroy(troyn_behz* behz);
uint32_t troyn_behz_base_Bsk_size(const troyn_behz* behz);
int troyn_behz_get_base_Bsk(const troyn_behz* behz, uint64_t* out); /* host copy, KAT hook: the base utils/rns_tool.cu:52-80 picks */
/* Number of primes of the auxiliary base troyn_bfv_multiply actually works in.  Equal to troyn_behz_base_Bsk_size unless every q_i is below
 * 2^50: then the multiply uses primes BELOW 2^50, sized by the reference's own criterion bits(prod(B) m_sk) > 32 + bits(t) + bits(q)
 * (utils/rns_tool.cu:50-62; results are independent of the choice), so that all of its transforms take the exact-FP64 butterflies
 * (plan option TROYN_BEHZ_BASE=ref at creation keeps the reference's base).  Sizes of intermediates follow this count. */
uint32_t troyn_behz_working_base_size(const troyn_behz* behz);
uint64_t troyn_behz_gamma(const troyn_behz* behz);
/* scaling_variant::scale_up / multiply_add_plain / multiply_sub_plain (utils/scaling_variant.cu:17-90,
 * fgk/translate_plain.cu:6-75): dest[item][L][N] = (from[item][L][N] or 0) +/- round(q/t * plain[item][i]);
 * coefficients i >= plain_coeff_count copy `from` (or 0).  Strides in elements; `from` may be NULL or == dest. */
int troyn_bfv_scale_up(const troyn_behz* behz, const uint64_t* plain, size_t plain_coeff_count, size_t plain_bstride,
                       const uint64_t* from, size_t from_bstride, uint64_t* dest, size_t dest_bstride,
                       int subtract, size_t batch, troyn_stream_t stream);
/* RNSTool::decrypt_scale_and_round (utils/rns_tool.cu:1189-1391): phase[batch][L][N] (c0 + c1*s + ..., coefficient
 * form) -> dest[batch][N] mod t. */
int troyn_bfv_decrypt_scale_and_round(const troyn_behz* behz, const uint64_t* phase, uint64_t* dest, size_t batch, troyn_stream_t stream);
size_t troyn_bfv_multiply_workspace_bytes(const troyn_behz* behz, size_t pa, size_t pb, size_t batch);
int troyn_bfv_multiply(const troyn_behz* behz, const uint64_t* a, size_t pa, const uint64_t* b, size_t pb,
                       uint64_t* out, void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TROYN_H */

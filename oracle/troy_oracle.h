/*
 * troy_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the *host* (CPU) algorithms of lightbulb128/troy-nova's
 * RNS-RLWE hot path: modular arithmetic, NTT tables, negacyclic NTT/INTT, RNS dyadic
 * products, key switching, modulus switching / CKKS rescale and the BEHZ BFV multiply.
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout's src/ directory).
 *
 * This library is the CHECKER.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (troy-nova_amd/) never links, imports or
 * calls anything in this directory; it fails loudly when its HIP library is missing.
 *
 * Parity pinning: the reference is CUDA (.cu sources that need nvcc + libcudart, neither
 * of which exists in this image), so it is UNBUILDABLE here and oracle/_ref is not
 * provided.  The oracle is pinned against the reference's own known-answer tests
 * (test/utils/ntt.cu, test/modulus.cu, test/utils/uint_small_mod.cu,
 * test/utils/rns_base.cu, test/utils/rns_tool.cu) re-typed as data in tests/golden/, and
 * against the values recorded in SURVEY.md (prime chains, BEHZ auxiliary primes), and
 * against OUTPUT of the reference itself: the digests of the BASELINE config-1 ciphertext
 * and of its BEHZ product that the reference's CPU branch produced for seed 0x123
 * (tests/golden/config1_digests.json, tests/test_oracle_config1.py).
 */
#ifndef TROY_ORACLE_H
#define TROY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar layer ------------------------------------------------------------------ */

/* modulus.h:8-124 (class Modulus) */
typedef struct {
    uint64_t value;
    uint64_t const_ratio[3]; /* floor(2^128/value) lo, hi ; 2^128 mod value */
    uint64_t bit_count;
    int32_t is_prime;
    int32_t pad_;
} orc_modulus;

/* utils/uint_small_mod.h:92-122 (struct MultiplyUint64Operand) */
typedef struct {
    uint64_t operand;
    uint64_t quotient;
} orc_mulop;

int orc_modulus_init(orc_modulus* m, uint64_t value); /* modulus.cu:7-32; -1 on invalid */
uint64_t orc_barrett_reduce64(uint64_t x, const orc_modulus* m);             /* modulus.h:22-42 */
uint64_t orc_barrett_reduce128(uint64_t lo, uint64_t hi, const orc_modulus* m); /* modulus.h:44-78 */
uint64_t orc_multiply_mod(uint64_t a, uint64_t b, const orc_modulus* m);    /* uint_small_mod.h:85-90 */
uint64_t orc_add_mod(uint64_t a, uint64_t b, const orc_modulus* m);         /* uint_small_mod.h:54-61 */
uint64_t orc_sub_mod(uint64_t a, uint64_t b, const orc_modulus* m);         /* uint_small_mod.h:64-72 */
uint64_t orc_negate_mod(uint64_t a, const orc_modulus* m);                  /* uint_small_mod.h:30-36 */
void orc_mulop_init(orc_mulop* o, uint64_t operand, const orc_modulus* m);   /* uint_small_mod.h:97-113 */
uint64_t orc_mulop_mod(uint64_t x, const orc_mulop* y, const orc_modulus* m);      /* :130-139 */
uint64_t orc_mulop_mod_lazy(uint64_t x, const orc_mulop* y, const orc_modulus* m); /* :142-148 */
uint64_t orc_exponentiate_mod(uint64_t a, uint64_t e, const orc_modulus* m);       /* :215-231 */
int orc_try_invert_mod(uint64_t a, const orc_modulus* m, uint64_t* out);           /* basics.h try_invert_uint64_mod */
uint64_t orc_dot_product_mod(const uint64_t* a, const uint64_t* b, size_t n, const orc_modulus* m); /* :251-261 */
int orc_is_prime(uint64_t value);                                             /* uint_small_mod.h:264-301 */

/* utils/number_theory.cu:22-39 ; returns number of primes written (== count) or -1 */
int orc_get_primes(uint64_t factor, size_t bit_size, size_t count, uint64_t* out);
/* coeff_modulus.cu:65-108 */
int orc_coeff_modulus_create(size_t poly_modulus_degree, const size_t* bit_sizes, size_t n, uint64_t* out);
/* utils/number_theory.cu:68-87 */
int orc_try_minimal_primitive_root(uint64_t degree, const orc_modulus* m, uint64_t* out);

/* ---- NTT --------------------------------------------------------------------------- */

/* utils/ntt.h:12-87, utils/ntt.cu:14-76 */
typedef struct {
    uint64_t root;
    size_t coeff_count_power;
    size_t coeff_count;
    orc_modulus modulus;
    orc_mulop inv_degree_modulo;
    orc_mulop* root_powers;     /* [coeff_count], bit-reversed order */
    orc_mulop* inv_root_powers; /* [coeff_count], scrambled order */
} orc_ntt_tables;

orc_ntt_tables* orc_ntt_tables_create(size_t coeff_count_power, uint64_t modulus);
void orc_ntt_tables_destroy(orc_ntt_tables* t);
uint64_t orc_ntt_tables_root(const orc_ntt_tables* t);
uint64_t orc_ntt_tables_root_power(const orc_ntt_tables* t, size_t i, int inverse, int want_quotient);
uint64_t orc_ntt_tables_inv_degree(const orc_ntt_tables* t, int want_quotient);

/* utils/ntt.h:89-130 NTTTableIndexer modes */
enum { ORC_IDX_COMPONENTWISE = 0, ORC_IDX_KS_SET_PRODUCTS = 1, ORC_IDX_KS_SKIP_FINALS = 2 };

/* fgk/ntt_grouped.cu:11-56 + :258-270 (host branch): in-place forward negacyclic NTT over
 * data[pcount][component_count][N]; tables = array of `n_tables` pointers. */
void orc_ntt_forward(uint64_t* data, size_t pcount, size_t component_count, size_t log_degree,
                     const orc_ntt_tables* const* tables, size_t n_tables, int indexer_mode, size_t decomp_size);
/* fgk/ntt_grouped.cu:346-391 + :597-610 (host branch) */
void orc_ntt_inverse(uint64_t* data, size_t pcount, size_t component_count, size_t log_degree,
                     const orc_ntt_tables* const* tables, size_t n_tables, int indexer_mode, size_t decomp_size);

/* ---- element-wise RNS polynomial ops (utils/poly_small_mod.cu host_* loops) ---------- */
void orc_add_ps(const uint64_t* a, const uint64_t* b, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_sub_ps(const uint64_t* a, const uint64_t* b, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_negate_ps(const uint64_t* a, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_modulo_ps(const uint64_t* a, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_multiply_uint64operand_ps(const uint64_t* a, const orc_mulop* operand, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_multiply_scalar_ps(const uint64_t* a, uint64_t scalar, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
void orc_dyadic_product_ps(const uint64_t* a, const uint64_t* b, size_t pcount, size_t degree, const orc_modulus* moduli, size_t nmod, uint64_t* out);
/* fgk/dyadic_convolute.cu:43-80 (host branch) / :116-140 */
void orc_dyadic_convolute(const uint64_t* a, const uint64_t* b, size_t pa, size_t pb, const orc_modulus* moduli, size_t nmod, size_t degree, uint64_t* out);
void orc_dyadic_square(const uint64_t* a, const orc_modulus* moduli, size_t nmod, size_t degree, uint64_t* out);

/* ---- RNS tool (utils/rns_base.cu, utils/rns_tool.cu) -------------------------------- */
typedef struct orc_rns_tool orc_rns_tool;
orc_rns_tool* orc_rns_tool_create(size_t poly_modulus_degree, const uint64_t* q, size_t q_size, uint64_t t);
void orc_rns_tool_destroy(orc_rns_tool* r);
size_t orc_rns_tool_base_B_size(const orc_rns_tool* r);
size_t orc_rns_tool_base_Bsk_size(const orc_rns_tool* r);
uint64_t orc_rns_tool_m_sk(const orc_rns_tool* r);
uint64_t orc_rns_tool_gamma(const orc_rns_tool* r);
uint64_t orc_rns_tool_m_tilde(const orc_rns_tool* r);
void orc_rns_tool_base_Bsk(const orc_rns_tool* r, uint64_t* out);
uint64_t orc_rns_tool_inv_q_last_mod_q(const orc_rns_tool* r, size_t i, int want_quotient);
/* generic BaseConverter::fast_convert_array, utils/rns_base.cu:350-380, exposed for KATs */
void orc_fast_convert_array(const uint64_t* ibase, size_t ni, const uint64_t* obase, size_t no,
                            const uint64_t* input, size_t count, uint64_t* output);
void orc_rns_fast_b_conv_m_tilde(const orc_rns_tool* r, const uint64_t* input, uint64_t* dest); /* rns_tool.cu:1083-1094 */
void orc_rns_sm_mrq(const orc_rns_tool* r, const uint64_t* input, uint64_t* dest);              /* :870-905 */
void orc_rns_fast_floor(const orc_rns_tool* r, const uint64_t* input, uint64_t* dest);          /* :973-988,:1010-1036 */
void orc_rns_fast_b_conv_sk(const orc_rns_tool* r, const uint64_t* input, uint64_t* dest);      /* :762-790,:831-868 */
void orc_rns_fast_b_conv_m_tilde_sm_mrq(const orc_rns_tool* r, const uint64_t* input, uint64_t* dest); /* :1096-1104 */
void orc_rns_fast_floor_fast_b_conv_sk(const orc_rns_tool* r, const uint64_t* in_q, const uint64_t* in_Bsk, size_t dest_size, uint64_t* dest); /* :1038-1075 */
/* rns_tool.cu:421-466 (host branch): in [pcount][L][N] -> out [pcount][L-1][N] */
void orc_rns_divide_and_round_q_last(const orc_rns_tool* r, const uint64_t* input, size_t pcount, uint64_t* dest);
/* rns_tool.cu:664-694 (host branch); tables = the level's L tables */
void orc_rns_divide_and_round_q_last_ntt(const orc_rns_tool* r, const uint64_t* input, size_t pcount, uint64_t* dest,
                                          const orc_ntt_tables* const* tables);

/* ---- context: modulus chain (he_context.cu:46-123, context_data.cu:71-345) ------------ */
enum { ORC_SCHEME_BFV = 1, ORC_SCHEME_CKKS = 2, ORC_SCHEME_BGV = 3 };
typedef struct orc_context orc_context;
/* coeff_modulus = full key-level chain (K primes, last = special prime when K > 1). */
orc_context* orc_context_create(int scheme, size_t poly_modulus_degree, const uint64_t* coeff_modulus, size_t K, uint64_t plain_modulus);
void orc_context_destroy(orc_context* c);
size_t orc_context_key_modulus_size(const orc_context* c);
const orc_ntt_tables* orc_context_ntt_table(const orc_context* c, size_t i);
/* level is addressed by its number of limbs `nlimbs` (K for the key level, K-1 for the first data level, ...) */
const orc_rns_tool* orc_context_rns_tool(const orc_context* c, size_t nlimbs);
const orc_modulus* orc_context_moduli(const orc_context* c);

/* evaluator_transform_ntt.cu:525-538 / :621-634 : ct[pcount][L][N] in place */
void orc_transform_to_ntt(const orc_context* c, uint64_t* ct, size_t pcount, size_t L);
void orc_transform_from_ntt(const orc_context* c, uint64_t* ct, size_t pcount, size_t L);

/* SwitchKeyDestinationAssignMethod, evaluator.h */
enum { ORC_ASSIGN_ADD_INPLACE = 0, ORC_ASSIGN_OVERWRITE = 1, ORC_ASSIGN_OVERWRITE_EXCEPT_FIRST = 2 };
/* evaluator_keyswitching_core.cu:757-1052 (host branches :833-902, :923-985).
 * target[L][N]; keys[j] (j < L) -> u64[2][K][N] NTT form; destination[2][L][N]. */
void orc_switch_key(const orc_context* c, size_t L, int is_ntt_form, const uint64_t* target,
                    const uint64_t* const* keys, int assign_method, uint64_t* destination);
/* evaluator_keyswitching.cu:96-144: ct[3][L][N] -> out[2][L][N] (relinearize_new) */
void orc_relinearize(const orc_context* c, size_t L, int is_ntt_form, const uint64_t* ct3, const uint64_t* const* keys, uint64_t* out2);
/* evaluator.cu:118-144 ckks_multiply : a[pa][L][N] x b[pb][L][N] -> out[pa+pb-1][L][N] */
void orc_ckks_multiply(const orc_context* c, size_t L, const uint64_t* a, size_t pa, const uint64_t* b, size_t pb, uint64_t* out);
/* evaluator.cu:29-116 bfv_multiply (BEHZ) */
void orc_bfv_multiply(const orc_context* c, size_t L, const uint64_t* a, size_t pa, const uint64_t* b, size_t pb, uint64_t* out);
/* evaluator_modswitch.cu:14-74 : BFV -> divide_and_round_q_last ; CKKS -> _ntt ; in[p][L][N] -> out[p][L-1][N] */
void orc_mod_switch_scale_to_next(const orc_context* c, size_t L, const uint64_t* in, size_t pcount, uint64_t* out);
/* evaluator_modswitch.cu:164-220 mod_switch_drop_to_next (CKKS mod_switch_to_next): drop last limb */
void orc_mod_switch_drop_to_next(const orc_context* c, size_t L, const uint64_t* in, size_t pcount, uint64_t* out);

/* ---- BASELINE config 1 host path (encode -> keygen -> encrypt), for pinning against the reference's digests ----
 * utils/random_generator.cu (AES-128-CTR PRNG :37-58,:238-247,:277-280; ternary :318-336; centered binomial :374-385,
 * :421-440; uniform :475-481), key_generator.cu:31-58 (secret key), :65-84 + utils/rlwe.cu:218-317 (public key),
 * batch_encoder.cu:14-64,:169-226 (encode), encryptor.cu:12-110,:259-268 + utils/rlwe.cu:11-91 (asymmetric encryption
 * with modulus switch from the key level), fgk/translate_plain.cu:28-38 (Delta scaling).  BFV only. */
typedef struct orc_rng orc_rng;
void orc_aes128_encrypt_block(uint8_t block[16], const uint8_t key[16]);
orc_rng* orc_rng_create(uint64_t seed_low, uint64_t seed_high);
void orc_rng_destroy(orc_rng* r);
uint64_t orc_rng_sample_uint64(orc_rng* r);
void orc_rng_fill_uint64s(orc_rng* r, uint64_t* out, size_t n);
void orc_sample_poly_ternary(orc_rng* r, uint64_t* dest, size_t degree, const orc_modulus* moduli, size_t nmod);
void orc_sample_poly_centered_binomial(orc_rng* r, uint64_t* dest, size_t degree, const orc_modulus* moduli, size_t nmod);
void orc_sample_poly_uniform(orc_rng* r, uint64_t* dest, size_t degree, const orc_modulus* moduli, size_t nmod);
/* secret key [K][N] (NTT form) and public key [2][K][N] (NTT form) drawn from `rng` in the reference's call order */
void orc_keygen_secret_key(const orc_context* c, orc_rng* rng, uint64_t* sk);
void orc_keygen_public_key(const orc_context* c, orc_rng* rng, const uint64_t* sk, uint64_t* pk);
/* BatchEncoder::encode: values[count] -> plain[N] mod t */
int orc_batch_encode(const orc_context* c, const uint64_t* values, size_t count, uint64_t* plain);
/* Encryptor::encrypt_asymmetric (BFV, plain at parms_id_zero): out [2][K-1][N], coefficient form */
void orc_encrypt_asymmetric_bfv(const orc_context* c, orc_rng* rng, const uint64_t* pk, const uint64_t* plain, size_t plain_coeff_count, uint64_t* out);
/* KeyGenerator::generate_rlk(1) (key_generator.cu:136-153,:206-237): L = K-1 keys, out [L][2][K][N] */
void orc_keygen_relin_keys(const orc_context* c, orc_rng* rng, const uint64_t* sk, uint64_t* out);
/* RNSTool::decrypt_scale_and_round, host branch (utils/rns_tool.cu:1118-1136,:1334-1370): phase [q_size][N] -> dest [N] mod t */
int orc_rns_decrypt_scale_and_round(const orc_rns_tool* r, const uint64_t* phase, uint64_t* dest);
/* Decryptor::bfv_decrypt (decryptor.cu:27-105,:268-362): ct [pcount][L][N] coefficient form, sk [K][N] NTT form -> plain [N] */
int orc_decrypt_bfv(const orc_context* c, const uint64_t* sk, const uint64_t* ct, size_t pcount, size_t L, uint64_t* plain);
/* BatchEncoder::decode (batch_encoder.cu): plain [N] mod t -> values [N] */
int orc_batch_decode(const orc_context* c, const uint64_t* plain, uint64_t* values);
/* ---- ciphertext x plaintext (BASELINE config 5 path) ----
 * scaling_variant::centralize, fast-plain-lift case (utils/scaling_variant.cu:258-275,:326-357): plain [count] mod t ->
 * dest [L][N] (coefficient form); returns -1 when some q_i <= t (no fast plain lift). */
int orc_plain_centralize(const orc_context* c, size_t L, const uint64_t* plain, size_t plain_coeff_count, uint64_t* dest);
/* Evaluator::multiply_plain_normal (evaluator_multiply_plain.cu:13-68): ct [pcount][L][N] coefficient form x plain (mod t,
 * parms_id_zero) -> out [pcount][L][N] coefficient form */
int orc_multiply_plain_normal(const orc_context* c, size_t L, const uint64_t* ct, size_t pcount, const uint64_t* plain, size_t plain_coeff_count, uint64_t* out);
/* Evaluator::multiply_plain_ntt (:196-218): NTT-form ct x NTT-form RNS plaintext [L][N] */
void orc_multiply_plain_ntt(const orc_context* c, size_t L, const uint64_t* ct, size_t pcount, const uint64_t* plain_ntt, uint64_t* out);
/* ---- Galois automorphisms ----
 * GaloisTool::apply_ps / apply_ntt_ps, host branches (utils/galois.cu:24-41,:147-166,:250-270): data [pcount][nmod][N] */
void orc_apply_galois(const orc_context* c, size_t nmod, int is_ntt_form, size_t galois_element, const uint64_t* in, size_t pcount, uint64_t* out);
/* Evaluator::apply_galois (evaluator_keyswitching.cu:147-179): ct [2][L][N]; keys = the L key-switching keys of the element */
void orc_apply_galois_ct(const orc_context* c, size_t L, int is_ntt_form, size_t galois_element, const uint64_t* ct,
                         const uint64_t* const* keys, uint64_t* out);
/* KeyGenerator::generate_galois_keys for ONE element (key_generator.cu:239-260): out [L][2][K][N] */
void orc_keygen_galois_key(const orc_context* c, orc_rng* rng, const uint64_t* sk, size_t galois_element, uint64_t* out);
/* GaloisTool::get_element_from_step (utils/galois.cu:43-63) */
size_t orc_galois_element_from_step(size_t n, int step);
/* ---- BGV.  A context created with ORC_SCHEME_BGV multiplies the key-generation and encryption
 * noise by t (utils/rlwe.cu:82-86,:300-304) and finishes orc_switch_key with the ski_util5 tail
 * (evaluator_keyswitching_core.cu:272-318).  BGV ciphertexts are NTT form; multiply = the dyadic product of
 * orc_ckks_multiply with correction factors multiplied mod t (evaluator.cu:150-173). ----
 * RNSTool::mod_t_and_divide_q_last_ntt (utils/rns_tool.cu:1540-1590): [pcount][nl][N] -> [pcount][nl-1][N]; the
 * ciphertext's correction factor is multiplied by orc_bgv_inv_q_last_mod_t (evaluator_modswitch.cu:70-72) */
void orc_rns_mod_t_and_divide_q_last_ntt(const orc_context* c, size_t nl, const uint64_t* input, size_t pcount, uint64_t* dest);
uint64_t orc_bgv_inv_q_last_mod_t(const orc_context* c, size_t nl);
/* RNSTool::decrypt_mod_t = BaseConverter::exact_convey_array (utils/rns_base.cu:445-529): phase [nl][N] -> [N] mod t */
int orc_rns_decrypt_mod_t(const orc_context* c, size_t nl, const uint64_t* phase, uint64_t* dest);
/* the same two on a bare RNSTool (any coprime base, no NTT tables needed: the shape of the reference's own known-answer tests, test/utils/rns_tool.cu:470-640):
 * RNSTool::decrypt_mod_t, phase [q_size][N] -> [N] mod t; RNSTool::mod_t_and_divide_q_last_inplace (utils/rns_tool.cu:1432-1466,:1515-1538), [q_size][N] in place */
int orc_rns_tool_decrypt_mod_t(const orc_rns_tool* r, const uint64_t* phase, uint64_t* dest);
void orc_rns_tool_mod_t_and_divide_q_last_inplace(const orc_rns_tool* r, uint64_t* input);
/* Decryptor::bgv_decrypt (decryptor.cu:509-539) */
int orc_decrypt_bgv(const orc_context* c, const uint64_t* sk, const uint64_t* ct, size_t pcount, size_t L, uint64_t correction_factor, uint64_t* plain);
/* Encryptor::encrypt_asymmetric for BGV, plaintext mod t (encryptor.cu:300-333): out [2][K-1][N] NTT form */
int orc_encrypt_asymmetric_bgv(const orc_context* c, orc_rng* rng, const uint64_t* pk, const uint64_t* plain, size_t plain_coeff_count, uint64_t* out);
/* ---- RLWE / LWE packing primitives (evaluator_lwes.cu) ----
 * utils::negacyclic_shift_ps host branch (utils/poly_small_mod.cu:902-925): data [pcount][nmod][N], shift in [0, 2N) */
void orc_negacyclic_shift(const orc_context* c, size_t nmod, const uint64_t* in, size_t pcount, size_t shift, uint64_t* out);
/* utils::ntt_multiply_inv_degree host branch (utils/ntt.cu:78-91): data * N^-1 * scalar, in place */
void orc_multiply_inv_degree(const orc_context* c, size_t nmod, uint64_t* data, size_t pcount, uint64_t scalar);
/* Evaluator::extract_lwe_new (evaluator_lwes.cu:52-97): ct [2][L][N] coefficient form -> c0 [L], c1 [L][N] */
void orc_extract_lwe(const orc_context* c, size_t L, const uint64_t* ct, size_t term, uint64_t* c0, uint64_t* c1);
/* the survey's digest: h = FNV offset; for each 64-bit WORD: h ^= word; h *= FNV prime */
uint64_t orc_fnv_words(const uint64_t* data, size_t n);

/* deterministic 64-bit generator used by tests/bench to make identical inputs on both sides
 * (splitmix64; NOT the reference's AES PRNG) ; fills out[i] uniformly in [0, bound) */
void orc_fill_uniform(uint64_t seed, uint64_t bound, uint64_t* out, size_t n);
/* FNV-1a-64 over the little-endian bytes of a u64 stream (SURVEY.md Appendix C digest) */
uint64_t orc_fnv1a64(const uint64_t* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif

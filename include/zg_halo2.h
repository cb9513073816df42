/*
 * zg_halo2.h -- C ABI of the MI355X proving backend for zero_g's Halo2 WNN circuit.
 *
 * This is the drop-in boundary.  The reference has no FFI seam of its own:
 * `Wnn::proof` (/root/reference/src/wnn.rs:232-262) calls `halo2_proofs::plonk::create_proof`
 * (wnn.rs:242-259) with `KZGCommitmentScheme<Bn256>` / `ProverGWC` (wnn.rs:243-244), and the hot
 * arithmetic sits behind halo2's `arithmetic::{best_multiexp, best_fft}`, `poly::EvaluationDomain`,
 * `poly::commitment::{Params, ParamsProver}` and `plonk::{evaluation, lookup, permutation,
 * vanishing}` (git dependency halo2_proofs v2023_04_20, reference Cargo.toml:21-25).  Each entry
 * point below names the upstream function it replaces; INTEGRATION.md shows the Rust `extern "C"`
 * binding a maintainer would add in a `[patch]`-ed halo2_proofs (the mechanism the reference
 * already uses for halo2curves, Cargo.toml:14-18).
 *
 * Conventions
 *   - zg_fr / zg_fq : 4 x u64 little-endian limbs, Montgomery form, R = 2^256 (halo2curves 0.3.3).
 *   - zg_g1_affine  : {x, y}, 64 B, (0,0) = identity (bn256::G1Affine).
 *   - zg_g1         : Jacobian {x, y, z}, 96 B, z = 0 = identity (bn256::G1).  MSM results are
 *                     returned NORMALISED (z = 1, or (0,1,0) for the identity) so that the bytes are
 *                     canonical: a Jacobian triple is otherwise only defined up to scaling.
 *   - every function returns 0 on success or a negative zg_status; zg_last_error() describes the
 *     most recent failure on the calling thread.  Nothing throws or aborts across the ABI.
 *     Upstream MSM/FFT are infallible apart from length asserts; the Rust shim maps nonzero to panic!.
 *   - pointers are borrowed for the duration of the call unless a handle is returned.
 *   - one zg_ctx drives one GPU (one process per GPU; multi-GPU composition is in INTEGRATION.md).
 *   - thread safety: every entry point that takes a zg_ctx (or a zg_prover / zg_bases created on one) holds that
 *     context's lock for the whole call, so calls on ONE context serialise and calls on DIFFERENT contexts run
 *     concurrently (e.g. from rayon workers, each with its own context).  Objects that are only read -- zg_bases
 *     tables, the proving key of forked provers -- may be shared by all contexts of their device.
 *   - *_dev entry points take DEVICE pointers (HBM-resident data, e.g. torch tensors' data_ptr())
 *     and are asynchronous on the context stream; call zg_ctx_sync() before reading results.
 */
#ifndef ZG_HALO2_H
#define ZG_HALO2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } zg_fr;
typedef struct { uint64_t l[4]; } zg_fq;
typedef struct { zg_fq x, y; } zg_g1_affine;
typedef struct { zg_fq x, y, z; } zg_g1;

typedef struct zg_ctx zg_ctx;       /* one GPU: stream, workspace, twiddle cache        */
typedef struct zg_bases zg_bases;   /* HBM-resident base set (ParamsKZG::g / g_lagrange) */

typedef enum {
    ZG_OK = 0,
    ZG_ERR_INVALID_ARG = -1,   /* null pointer, length mismatch (upstream: assert_eq! panic) */
    ZG_ERR_NO_DEVICE = -2,     /* no usable HIP device: the library never falls back to a CPU path */
    ZG_ERR_HIP = -3,           /* a HIP runtime call failed                                  */
    ZG_ERR_UNSUPPORTED = -4,   /* size outside what the kernels are built for                */
    ZG_ERR_CONSTRAINT = -5,    /* plonk::Error::ConstraintSystemFailure (lookup input not in table) */
    ZG_ERR_OOM = -6
} zg_status;

const char *zg_last_error(void);
/* Library version string and the gfx target it was compiled for ("gfx950"). */
const char *zg_version(void);

/* ------------------------------------------------------------------ tuning
 * Every switch of the library, by name.  A knob starts from the environment variable of the same name (read once, when
 * the first knob is asked for) and may be set at run time; value < 0 restores the library's default.  None changes a
 * result: every form computes the same group elements and field elements, bit for bit (tests/test_gpu_knobs.py walks
 * them all against the oracle).  Knobs that shape resident data are read when that object is built (ZG_MSM_C: a base
 * set registered with window_bits = 0; ZG_MSM_NAF, ZG_MSM_NAF_GL, ZG_MSM_RUNS, ZG_EVALH9, ZG_EVALH_GROUPED,
 * ZG_SPLIT_DOMAIN: zg_prover_create*; ZG_LAT_SPLIT_K: zg_prover_create* and zg_prover_set_overlap; ZG_LAT_FULL_C: zg_prover_enable_digit_tables), launch shapes at every launch (ZG_MSM_K, ZG_MSM_K_LAT, ZG_MSM_RB, ZG_MSM_LANES,
 * ZG_MSM_STRIP, ZG_LAT_FULL_K, ZG_LAZY_DOT, ZG_MSM_AFFINE, ZG_MSM_HEAVY, ZG_LAT_PULL, ZG_LAT_GATE).
 *   ZG_MSM_C          window bits of the MSM tables, 2..16 (default from n: k - 2)
 *   ZG_MSM_K          points per bucket-accumulation task in the throughput form, 4..120 (48)
 *   ZG_MSM_K_LAT      ... in the latency form (16; 32 / 48 from n = 2^16 / 2^17)
 *   ZG_MSM_RB         buckets per block of the latency form's bucket reduction: 64 / 128 / 256
 *   ZG_MSM_LANES      lanes per EC addition there: 2 / 4
 *   ZG_MSM_STRIP      buckets per lane in the throughput form's reduction: 2 / 4 / 8 / 16 (8)
 *   ZG_MSM_NAF        digit width of the free-position form for all-random commitments, 3..16; 0 = window tables only
 *   ZG_MSM_NAF_GL     ... for the run-form commitments against g_lagrange (c + 1); 0 = windows
 *   ZG_MSM_RUNS       0 = no run form (summation by parts over running base sums)
 *   ZG_EVALH_GROUPED  0 = evaluate_h folds in y term by term (the fallback of circuits with > 40 such terms)
 *   ZG_EVALH9         0 = evaluate_h on 8 x 32-bit limbs; implies the single extended coset
 *   ZG_SPLIT_DOMAIN   0 = EvaluationDomain's single extended coset in the throughput form too
 *   ZG_LAT_SPLIT_K    smallest k at which a lone proof (the latency form) takes the quotient from the split domain too (15)
 *   ZG_LAT_FULL_C     window bits of the latency form's digit tables (zg_bases_enable_digit_table); 0 = none
 *   ZG_LAT_FULL_K     summands per lane pair in the digit-table accumulation, 4..120 (16)
 *   ZG_LAZY_DOT       0 = eval_polynomial's dot products and the multiopen combinations reduce after every term
 *   ZG_MSM_AFFINE     rounds (0..4) of batched-affine pairwise additions -- one shared inversion per launch and round --
 *                     before the buckets' XYZZ chains in the throughput form; 0 = none
 *   ZG_MSM_HEAVY      task partials of a bucket above which it is merged ahead of the bucket reduction (msm_heavy), 1..64
 *                     (4 in the throughput form, 16 in the latency form)
 *   ZG_LAT_PULL       1 = a lone proof's per-phase scalars (challenges, opening points) reach the device through a one-wave
 *                     kernel that reads the pinned staging arena, 0 = through a copy command
 *   ZG_LAT_GATE       1 = a lone proof (latency form, one proof per call, unsharded; from the second proof in that form on)
 *                     puts each phase's launches on the stream BEFORE the host has the challenge they depend on, behind a
 *                     one-workgroup kernel that polls a word of mapped host memory; the host opens it with one store once
 *                     the challenge is staged (-1 to -2 % latency).  0 (default) = every phase is launched after its
 *                     challenge.  OPT-IN because a kernel that waits for the host needs every stream of the process on a
 *                     hardware queue of its own (GPU_MAX_HW_QUEUES >= the process's streams; two per latency-form prover):
 *                     where streams share a queue another stream's work can stand between this prover's streams and the
 *                     gate then only opens at its time limit (4 s), after which the proof is made again in the plain
 *                     order -- a stall, never a wrong proof.  Ignored under AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING; must be
 *                     OFF under counter collection (rocprofv3 --pmc serialises kernels across queues, which the library
 *                     cannot see).  A proof is gated only when it repeats the last completed proof's signature (scheduling
 *                     form, batch, instance length, digit tables, and no zg_tuning_set / zg_prover_set_batch in between):
 *                     anything else may allocate or synchronise and is made in the plain order; should a queued-ahead
 *                     phase still reach a blocking call, the gate is let go at once and the proof re-made (zg_prover_gate_stats).
 *                     (2: as 1, with the first gate of every proof left closed for 0.2 s -- exercises that path in tests.)
 *   ZG_WITNESS_LDS    0 = a witness plan (zg_witness_plan_create) keeps every operand of its program in HBM; 1 (default) =
 *                     the plan is register-allocated into LDS cells when it is made (live values in LDS, a level costs an LDS
 *                     round trip instead of three HBM round trips); n >= 2 = at most n KB of cells (what does not fit stays
 *                     in HBM: the medium and large models' case at the full 150 KB); read when the plan is created
 *   ZG_MSM_TOPSPLIT   0 = the free-position recoding (ZG_MSM_NAF) leaves its last digit whatever bits remain above the previous one
 *                     (tiny with probability 1/5: the hot small buckets); 1 (default) = the last two digits are cut evenly
 *   ZG_NTT9           the transforms' butterflies: 0 = on 8 x 32-bit limbs (a canonical value after every operation), 1 = on nine
 *                     29-bit limbs (limb-wise sums, no carry word in the products: a quarter fewer instructions per pass, 27 %
 *                     more multiply-adds; 36-byte elements in LDS).  Default: nine in the latency form (a lone proof: -0.3 / -1.7 /
 *                     -1.2 % at k = 14 / 15 / 17), eight in the throughput form (the nine-limb pass measured +0.4 ... +0.6 % ms/proof
 *                     under twelve provers: DESIGN.md section 5).  Same bytes either way: what leaves a pass is canonical. */
int zg_tuning_set(const char *name, int value);
int zg_tuning_get(const char *name, int *value);
/* out[i] = name of knob i for i < min(cap, count); returns the count. */
size_t zg_tuning_names(const char **out, size_t cap);

/* ------------------------------------------------------------------ context */
/* Creates a context on HIP device `device_id`.  Fails with ZG_ERR_NO_DEVICE when no GPU is
 * visible: there is deliberately no CPU fallback.  (SURVEY.md's sketch had zg_ctx_create(n_devices, device_ids):
 * here a context is ONE device -- one process per GPU -- and several GPUs are several contexts, composed by the
 * caller through zg_prover_set_shard / whole proofs per GPU.) */
int zg_ctx_create(int device_id, zg_ctx **out);
void zg_ctx_destroy(zg_ctx *ctx);
int zg_ctx_sync(zg_ctx *ctx);
/* Returns the free blocks of the context's workspace pool (and of its side stream's) to the device allocator, after
 * draining both streams; the pool grows again on demand.  For a long-lived process between workloads of different sizes
 * (upstream has no counterpart: halo2's buffers are Vec<F> owned by create_proof).  *freed_bytes (may be NULL) = what went. */
int zg_ctx_trim(zg_ctx *ctx, uint64_t *freed_bytes);
/* The hipStream_t every call of this context is ordered on (as void* to keep HIP out of the ABI). */
void *zg_ctx_stream(zg_ctx *ctx);

/* Per-kernel timing with HIP events recorded on the context stream around every launch (used by
 * bench.py for the roofline line; off by default).  Two ALGORITHMIC byte counts, neither measured HBM traffic:
 * algo_bytes = what the kernel's own algorithm streams (every distinct input once, every output once);
 * unit_bytes = SURVEY.md 8d's figure for the unit of work (one MSM: n * 96 + 96; one transform: (in + out) * 32; one
 * grand product: 3 n * 32; evaluate_h: (inputs + 1) * 8n * 32), charged ONCE per unit -- on the one kernel of a
 * multi-kernel launch sequence that carries the unit (msm_accumulate for an MSM), 0 on its stage kernels and on
 * kernels SURVEY names no unit for.  Summed over a proof's launches unit_bytes is SURVEY's per-proof figure. */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
    double algo_bytes;
    double unit_bytes;
} zg_kernel_stat;
int zg_ctx_profile_enable(zg_ctx *ctx, int on);
/* Time only launches of the named kernel (NULL or "" = every kernel).  A timed launch carries its own start and
 * stop event (hipExtLaunchKernelGGL): the kernel's begin and end as rocprofv3 reports them; two events per launch
 * are cheap for one kernel, not for the ~150 launches of a whole proof. */
int zg_ctx_profile_filter(zg_ctx *ctx, const char *kernel_name);
int zg_ctx_profile_collect(zg_ctx *ctx, zg_kernel_stat *out, size_t cap, size_t *count);

/* ------------------------------------------------------------------ MSM
 * Replaces halo2_proofs::arithmetic::best_multiexp as reached through
 * ParamsKZG::commit / commit_lagrange (halo2_proofs/src/poly/kzg/commitment.rs upstream),
 * i.e. every commitment create_proof makes (30 per WNN proof).        */

/* Uploads a fixed base set once (ParamsKZG::g or ::g_lagrange) and precomputes the
 * window-shifted copies 2^(c*w) * P_i that let all Pippenger windows share one bucket set.
 * window_bits = 0 picks c from n.  Host pointer. */
int zg_bases_register(zg_ctx *ctx, const zg_g1_affine *bases, size_t n, uint32_t window_bits,
                      zg_bases **out);
/* Same, from a device pointer (n * 64 B, HBM resident). */
int zg_bases_register_dev(zg_ctx *ctx, const void *d_bases, size_t n, uint32_t window_bits,
                          zg_bases **out);
void zg_bases_free(zg_bases *b);
/* Optional second table of a base set for batches of FULL-SIZE (random) scalars: one row per bit position, 2^j * P_i for
 * j < 255 (255 * n * 64 B -- 0.27 GB at n = 2^14), against which a scalar is recoded into odd signed digits of
 * `digit_width` bits at free positions (width-w NAF): 254 / (w + 1) additions per scalar where c-bit windows spend
 * 255 / c, on 2^(w-2) buckets.  Used by the zg_msm* entries of a context in its THROUGHPUT form
 * (zg_ctx_set_msm_latency(ctx, 0)); a latency-form context keeps the window table (a lone MSM waits on the gathers from
 * a table no cache holds).  digit_width in [3, 16]; idempotent for the same width.  Same results, bit for bit. */
int zg_bases_enable_bit_table(zg_ctx *ctx, zg_bases *bases, uint32_t digit_width);
/* Optional digit tables of a base set for LONE MSMs (a context in its latency form, zg_ctx_set_msm_latency(ctx, 1)): every
 * multiple d * 2^(c w) * P_i, d <= 2^(c-1), of every c-bit window -- ceil(255 / c) * 2^(c-1) * n * 64 B (26 GB at n = 2^14,
 * c = 11).  A signed digit then names its summand and the MSM is a flat sum of gathered points folded by a tree: no digit
 * sort, no buckets, no bucket reduction -- the latency of a lone commitment phase drops by about half.  window_bits 0 = chosen
 * from n and the free memory (none from n = 2^16 on, or when the card has no room: not an error); 4..12 otherwise.
 * Never built implicitly: this call for a lone base set, zg_prover_enable_digit_tables for a prover's three.
 * Same results, bit for bit. */
int zg_bases_enable_digit_table(zg_ctx *ctx, zg_bases *bases, uint32_t window_bits);
size_t zg_bases_len(const zg_bases *b);
uint32_t zg_bases_window_bits(const zg_bases *b);

/* Form of the MSM bucket reduction on this context: latency = 1 (default) spends two lanes per EC addition
 * (four in the suffix scan) -- shortest dependent chain, for a lone MSM or proof; latency = 0 spends one lane per addition, the
 * form for many MSMs in flight.  Results are identical.
 * zg_prover_set_overlap sets it together with the prover's own scheduling. */
int zg_ctx_set_msm_latency(zg_ctx *ctx, int latency);

/* A base set may be used from any context of the device it was registered on (the tables are read-only); it must
 * outlive every prover created on it, and zg_bases_free waits for the device to go idle. */
/* out = sum_i scalars[i] * bases[i], i < n <= zg_bases_len; == best_multiexp(scalars, &bases[..n]) */
int zg_msm(zg_ctx *ctx, const zg_bases *bases, const zg_fr *scalars, size_t n, zg_g1 *out);
/* `batch` scalar vectors against the same bases in one launch sequence; out[batch]. */
int zg_msm_batch(zg_ctx *ctx, const zg_bases *bases, const zg_fr *const *scalars, size_t batch,
                 size_t n, zg_g1 *out);
/* Device-resident scalars: vector b starts at d_scalars + b*stride_elems*32 B.  The un-normalised
 * results (extended Jacobian X,Y,ZZ,ZZZ; 128 B each) are left in d_out_xyzz[batch]; use
 * zg_msm_finish to bring them to the host as normalised zg_g1. */
int zg_msm_batch_dev(zg_ctx *ctx, const zg_bases *bases, const void *d_scalars, size_t stride_elems,
                     size_t batch, size_t n, void *d_out_xyzz);
int zg_msm_finish(zg_ctx *ctx, const void *d_xyzz, size_t batch, zg_g1 *out);
/* Host helper for point-range sharding across GPUs: out = normalised sum of `count` Jacobian
 * partials (the EC add that follows the RCCL all-gather; EC add is not an ncclRedOp). */
int zg_g1_sum(const zg_g1 *parts, size_t count, zg_g1 *out);

/* ------------------------------------------------------------------ SRS
 * Replaces ParamsKZG::<Bn256>::new(k) (reference call sites benches/bench.rs:19, src/main.rs:232):
 * g[i] = s^i * G and g_lagrange[i] = L_i(s) * G, 2^k affine points each.  Upstream draws s from
 * OsRng; here the caller supplies it so that runs are reproducible. */
int zg_params_new(zg_ctx *ctx, uint32_t k, const zg_fr *s, zg_g1_affine *g, zg_g1_affine *g_lagrange);
int zg_params_new_dev(zg_ctx *ctx, uint32_t k, const zg_fr *s, void *d_g, void *d_g_lagrange);

/* ------------------------------------------------------------------ NTT
 * Replaces halo2_proofs::arithmetic::best_fft and the EvaluationDomain wrappers
 * (halo2_proofs/src/poly/domain.rs upstream: lagrange_to_coeff, coeff_to_extended,
 * extended_to_coeff, divide_by_vanishing_poly).                                             */

/* In place, natural order in and out: a[k] <- sum_j a[j] * omega^(j*k); a has 2^log_n entries. */
int zg_ntt(zg_ctx *ctx, zg_fr *a, uint32_t log_n, const zg_fr *omega);
/* EvaluationDomain::ifft: best_fft with omega_inv followed by the multiplication by `divisor`. */
int zg_intt(zg_ctx *ctx, zg_fr *a, uint32_t log_n, const zg_fr *omega_inv, const zg_fr *divisor);
int zg_ntt_batch(zg_ctx *ctx, zg_fr *const *a, size_t batch, uint32_t log_n, const zg_fr *omega);
int zg_intt_batch(zg_ctx *ctx, zg_fr *const *a, size_t batch, uint32_t log_n,
                  const zg_fr *omega_inv, const zg_fr *divisor);
/* Device-resident, in place; array b at d_a + b*stride_elems*32 B.  divisor may be NULL. */
int zg_ntt_batch_dev(zg_ctx *ctx, void *d_a, size_t stride_elems, size_t batch, uint32_t log_n,
                     const zg_fr *omega, const zg_fr *divisor);

/* EvaluationDomain::coeff_to_extended: coefficient i is multiplied by zeta^(i mod 3)
 * (g_coset = Fr::ZETA), zero-padded from 2^k to 2^ext_k and transformed with extended_omega. */
int zg_coeff_to_extended(zg_ctx *ctx, const zg_fr *coeffs, uint32_t k, uint32_t ext_k, zg_fr *out);
int zg_coeff_to_extended_batch_dev(zg_ctx *ctx, const void *d_coeffs, size_t in_stride_elems,
                                   void *d_out, size_t out_stride_elems, size_t batch, uint32_t k,
                                   uint32_t ext_k);
/* EvaluationDomain::extended_to_coeff: inverse transform on the extended domain, multiplication by
 * 2^-ext_k and by zeta^-(i mod 3), truncated to out_len (= n * quotient_poly_degree). evals is
 * clobbered. */
int zg_extended_to_coeff(zg_ctx *ctx, zg_fr *evals, uint32_t k, uint32_t ext_k, size_t out_len,
                         zg_fr *out);
int zg_extended_to_coeff_dev(zg_ctx *ctx, void *d_evals, uint32_t k, uint32_t ext_k, size_t out_len,
                             void *d_out);

/* The standard domain roots: omega = ROOT_OF_UNITY^(2^(28-log_n)) and its inverse. */
int zg_domain_omega(uint32_t log_n, zg_fr *omega, zg_fr *omega_inv);

/* ------------------------------------------------------------------ circuit description
 * What halo2 keeps in `ConstraintSystem` + `ProvingKey` (halo2_proofs v2023_04_20
 * src/plonk/circuit.rs, src/plonk/keygen.rs), flattened to plain arrays so that it can cross a C ABI.
 * For zero_g it is produced from `WnnCircuit::configure` (/root/reference/src/gadgets/wnn.rs:334-371,
 * chips configured at wnn.rs:125-172); SURVEY.md appendix A lists its 12 gate polynomials,
 * 4 lookups and 8 permutation columns.  Every polynomial (gate, lookup input, lookup table) is given
 * in expanded form sum_m coeff_m * prod_f cell(query_f): exact over the field, so the value is the
 * one halo2's Expression tree / GraphEvaluator computes. */
enum { ZG_FIXED = 0, ZG_ADVICE = 1, ZG_INSTANCE = 2 };
enum { ZG_MAX_FACTORS = 8, ZG_MAX_LOOKUP_WIDTH = 4 };

typedef struct {
    uint32_t kind;      /* ZG_FIXED / ZG_ADVICE / ZG_INSTANCE */
    uint32_t column;
    int32_t rotation;   /* Rotation(i32): row offset in the 2^k domain */
} zg_query;

typedef struct {
    zg_fr coeff;                          /* Montgomery form */
    uint32_t n_factors;                   /* 0 = constant term */
    uint32_t factors[ZG_MAX_FACTORS];     /* indices into zg_circuit.queries */
} zg_monomial;

typedef struct { uint32_t first, count; } zg_poly;   /* range in zg_circuit.monomials */

typedef struct {
    uint32_t width;                              /* tuple length m (same for inputs and table) */
    zg_poly inputs[ZG_MAX_LOOKUP_WIDTH];         /* lookup::Argument::input_expressions */
    zg_poly tables[ZG_MAX_LOOKUP_WIDTH];         /* lookup::Argument::table_expressions */
} zg_lookup;

typedef struct {
    uint32_t k;                  /* rows = 2^k */
    uint32_t cs_degree;          /* ConstraintSystem::degree(); extended domain = smallest 2^e >= 2^k*(degree-1) */
    uint32_t blinding_factors;   /* ConstraintSystem::blinding_factors(); last rotation = -(bf+1) */
    uint32_t n_fixed, n_advice, n_instance;
    uint32_t n_queries;        const zg_query *queries;
    uint32_t n_monomials;      const zg_monomial *monomials;
    uint32_t n_gates;          const zg_poly *gates;         /* gate polynomials in creation order */
    uint32_t n_lookups;        const zg_lookup *lookups;
    uint32_t n_perm_columns;   const zg_query *perm_columns; /* permutation::Argument::columns (rotation unused) */
    uint32_t n_advice_queries; const zg_query *advice_queries;  /* cs.advice_queries: order of the advice evals */
    uint32_t n_fixed_queries;  const zg_query *fixed_queries;   /* cs.fixed_queries  */
} zg_circuit;

/* ------------------------------------------------------------------ prover
 * Replaces halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC, _, _,
 * EvmTranscript, _> for ONE circuit instance, as Wnn::proof calls it
 * (/root/reference/src/wnn.rs:242-259), from "advice columns assigned" to "proof bytes":
 * commitments (MSM), lagrange_to_coeff / coeff_to_extended (NTT), lookup::prover::{commit_permuted,
 * commit_product}, permutation::prover::commit, vanishing::prover::{commit, construct, evaluate},
 * Evaluator::evaluate_h, eval_polynomial and ProverGWC::create_proof, with the Keccak-256
 * EvmTranscript of snark-verifier (wnn.rs:21,249).  Witness synthesis (the Rust WnnChip) stays on the
 * caller's side: advice arrives as column values.
 *
 * Randomness: upstream draws every blinding scalar from OsRng (wnn.rs:256).  Here each one is
 * rand_fr(key, purpose tag, index): the ChaCha20 (RFC 7539) keystream under a 32-byte key supplied by the caller,
 * nonce = (tag, index), rejection-sampled below r.  A caller who passes 32 fresh bytes from its own
 * CSPRNG (the Rust shim: OsRng.fill_bytes) gets blinding that is as hidden as that key; a fixed key makes the proof
 * reproducible, which is what lets the oracle produce the same bytes -- tests use small integers as keys
 * (little-endian, zero-padded), production callers must not. */
typedef struct zg_prover zg_prover;

/* Uploads the proving key material and derives what keygen_pk derives (fixed/sigma polys and their
 * extended cosets, l_0 / l_last / l_active_row cosets).  fixed_values: [n_fixed][2^k] Lagrange values;
 * sigma_values: [n_perm_columns][2^k] Lagrange values of the permutation polynomials
 * (pk.permutation.permutations); g / g_lagrange: ParamsKZG; vk_repr: vk.transcript_repr, the scalar
 * create_proof hashes into the transcript first.  All host pointers, copied. */
int zg_prover_create(zg_ctx *ctx, const zg_circuit *circuit, const zg_fr *fixed_values,
                     const zg_fr *sigma_values, const zg_g1_affine *g, const zg_g1_affine *g_lagrange,
                     const zg_fr *vk_repr, zg_prover **out);
/* Same, with base tables registered once per device (zg_bases_register on any context of that device)
 * and shared read-only by several provers / proof streams: the window tables are the largest resident
 * object (2 * W * 2^k * 64 B) and sharing them keeps them in the Infinity Cache.  The bases must outlive
 * the prover and use the same window size.  Base sets with FEWER than 2^k points are a point-range shard of the
 * SRS: see zg_prover_set_shard. */
int zg_prover_create_shared(zg_ctx *ctx, const zg_circuit *circuit, const zg_fr *fixed_values,
                            const zg_fr *sigma_values, const zg_bases *g, const zg_bases *g_lagrange,
                            const zg_fr *vk_repr, zg_prover **out);
/* A second prover on the SAME proving key (no copy: the key's HBM is shared and freed with its last prover) and
 * the same base tables (tables the parent registered itself are owned jointly and freed with the last prover; tables
 * the caller registered stay the caller's), on another context of the device -- another stream of proofs or proof
 * batches. */
int zg_prover_fork(const zg_prover *parent, zg_ctx *ctx, zg_prover **out);
void zg_prover_destroy(zg_prover *p);

/* Lock-step batches.  A prover has `max_batch` proof slots (1 after creation); zg_prover_prove_batch makes up to
 * that many proofs of the same circuit AT ONCE: every kernel launch of create_proof serves all of them (the
 * commitments of a phase are one MSM over batch x columns vectors, the transforms one NTT batch, evaluate_h one
 * grid with a row of workgroups per proof ...), while each proof keeps its own transcript, challenges and
 * blinding key.  Proof bytes are those of the one-at-a-time calls. */
int zg_prover_set_batch(zg_prover *p, size_t max_batch);
size_t zg_prover_batch(const zg_prover *p);
/* Device address of slot `slot`'s advice columns, [n_advice][2^k] (a witness generator may write there directly);
 * valid until the next zg_prover_set_batch. */
void *zg_prover_advice_slot(zg_prover *p, size_t slot);
/* count <= max_batch proofs.  advice[b]: [n_advice][2^k] column values of proof b (host), or NULL when the slot
 * already holds them; the last blinding_factors+1 rows are overwritten with blinding scalars as create_proof does.
 * instance[b]: [n_instance][instance_len].  rng_keys: count x 32 bytes.  proofs[b] (proof_cap bytes each) receives
 * proof b's transcript bytes, proof_lens[b] their length (0 on failure), statuses[b] (optional) its zg_status: a
 * witness that fails a lookup fails ITS proof (ZG_ERR_CONSTRAINT), the others complete.  Returns the first
 * non-zero status. */
int zg_prover_prove_batch(zg_prover *p, size_t count, const zg_fr *const *advice, const zg_fr *const *instance,
                          size_t instance_len, const uint8_t *rng_keys, uint8_t *const *proofs, size_t proof_cap,
                          size_t *proof_lens, int *statuses);
/* Same with the advice columns in HBM: d_advice[b] is a device pointer (copied into the slot unless it IS the slot,
 * zg_prover_advice_slot) or NULL = the slot as it stands; d_advice itself may be NULL. */
int zg_prover_prove_batch_dev(zg_prover *p, size_t count, void *const *d_advice, const zg_fr *const *instance,
                              size_t instance_len, const uint8_t *rng_keys, uint8_t *const *proofs,
                              size_t proof_cap, size_t *proof_lens, int *statuses);
/* One proof (the batch of one).  advice: [n_advice][2^k] column values (host).  instance: [n_instance][instance_len]
 * public inputs.  proof: receives the transcript bytes (EvmTranscript layout). */
int zg_prover_prove(zg_prover *p, const zg_fr *advice, const zg_fr *instance, size_t instance_len,
                    const uint8_t rng_key[32], uint8_t *proof, size_t proof_cap, size_t *proof_len);
/* Same with the advice columns already in HBM ([n_advice][2^k]; copied into slot 0 unless they are slot 0). */
int zg_prover_prove_dev(zg_prover *p, void *d_advice, const zg_fr *instance, size_t instance_len,
                        const uint8_t rng_key[32], uint8_t *proof, size_t proof_cap, size_t *proof_len);

/* Point-range shard of the commitments across `world` GPUs (one process and one prover per GPU):
 * this prover's base sets hold points [first_point, first_point + zg_bases_len) of ParamsKZG::g / ::g_lagrange, it
 * multiplies only that range of every scalar vector, and per commitment phase the ranks exchange their partial sums
 * -- `exchange` is an ALL-GATHER: `bytes` bytes of `send` from every rank, rank order, into recv[world * bytes]
 * (torch.distributed / RCCL on the caller's side; EC addition is not a reduction operator, so it is a gather plus
 * local additions, never an all-reduce).  Every rank runs the rest of create_proof (transforms, evaluate_h) itself and
 * ends with the same proof bytes. */
typedef int (*zg_exchange_fn)(void *user, const void *send, size_t bytes, void *recv);
/* Host helper, the additions behind that all-gather: parts = world x count partial sums in the MSM's extended
 * Jacobian form (X, Y, ZZ, ZZZ; 128 B each, as zg_msm_batch_dev leaves them), out[i] = normalised sum over the ranks. */
int zg_xyzz_sum_ranks(const void *parts, size_t world, size_t count, zg_g1 *out);
int zg_prover_set_shard(zg_prover *p, uint32_t rank, uint32_t world, size_t first_point, zg_exchange_fn exchange,
                        void *user);
/* The same shard with the exchange INSIDE the library: nccl_comm is an initialised RCCL communicator (ncclComm_t) of
 * `world` ranks whose rank order is the point-range order, one per prover (a communicator serialises its collectives
 * on one stream; forks do not inherit it).  Per commitment phase: ONE ncclAllGather of the phase's partial sums
 * (128 B each) over xGMI on the prover's stream, the world - 1 additions per commitment in a kernel behind it, and
 * only whole sums reach the host -- north_star's "single RCCL reduce of partial sums", spelt as gather + additions
 * because EC addition is no ncclRedOp.  librccl.so is bound with dlopen when this entry is first used. */
int zg_prover_set_shard_rccl(zg_prover *p, uint32_t rank, uint32_t world, size_t first_point, void *nccl_comm);
/* The additions of that path as an entry of their own (device pointers, on the context's stream, not normalised:
 * d_out[i] = sum over r of d_parts[r * count + i], both in the 128-byte extended Jacobian form). */
int zg_xyzz_sum_ranks_dev(zg_ctx *ctx, const void *d_parts, size_t world, size_t count, void *d_out);
/* Upper bound of the proof size in bytes for this circuit. */
size_t zg_prover_proof_size(const zg_prover *p);
/* Test hook: copies an intermediate of the LAST proof to the host.  what: 0 = h(X) on the extended
 * coset after the division by (X^n - 1) [2^ext_k], 1 = permutation z (index = set) [2^k Lagrange],
 * 2 = lookup z (index = lookup) [2^k Lagrange], 3 = permuted input a' (index = lookup),
 * 4 = permuted table s' (index = lookup), 5 = h pieces in coefficient form [5 * 2^k]. */
int zg_prover_fetch(zg_prover *p, uint32_t what, uint32_t index, zg_fr *out, size_t cap_elems);
/* ... of proof `slot` of the last batch. */
int zg_prover_fetch_slot(zg_prover *p, size_t slot, uint32_t what, uint32_t index, zg_fr *out, size_t cap_elems);

/* Host wall-clock milliseconds the LAST proof spent per phase (diagnostics): 0 instance+advice
 * commitments, 1 lookup compression + permutation (of which out[7] is the host sort) + commitments,
 * 2 permutation/lookup products + random poly, 3 evaluate_h + h commitments, 4 evaluations,
 * 5 GWC openings, 6 total. */
int zg_prover_phase_ms(const zg_prover *p, double *out, size_t cap);
/* What the gate (ZG_LAT_GATE) did on this prover so far (diagnostics; upstream has no counterpart -- create_proof there
 * launches nothing ahead): out[0] proofs made in the gated order, out[1] gates armed, out[2] proofs re-made in the plain
 * order because a gate ran into its time limit or was let go, out[3] gates let go early by a blocking call. */
int zg_prover_gate_stats(const zg_prover *p, uint64_t *out, size_t cap);

/* Scheduling of one proof on its GPU.  enable = 1 (default): the coefficient / coset transforms run on a
 * second HIP stream beside the commitment MSMs -- lowest latency for a lone proof.  enable = 0: one
 * stream per proof -- the throughput configuration when many provers share the GPU (other proofs fill
 * the gaps, and every extra stream costs a hardware queue); in this form the quotient is also computed on a
 * split extended domain (4n + n points instead of EvaluationDomain's 8n for a degree-6 circuit: the same
 * polynomial h from fewer evaluations) and the MSM reductions spend one lane per EC addition.  Proof bytes do
 * not depend on it. */
int zg_prover_set_overlap(zg_prover *p, int enable);
/* Builds the digit tables (zg_bases_enable_digit_table) of the prover's three base sets -- ParamsKZG::g, ::g_lagrange and
 * the running sums of g_lagrange -- which a LONE proof (zg_prover_set_overlap(p, 1)) then uses instead of Pippenger buckets:
 * 2.9 -> 2.2 ms per create_proof at k = 14.  Footprint: 3 x ceil(255 / c) * 2^(c-1) * n * 64 B -- 78 GB at n = 2^14 (c = 11),
 * 84 GB at n = 2^15 (c = 10) -- and ~0.3 s of build time per table (0.8 s for the three at n = 2^14), which is why it is an
 * explicit call (round 3 built them inside the first latency-form proof, in 3.7 s).  max_bytes = what the three tables may take together; 0 = the library's cap (a third
 * of the card's memory, at most 90 GB) -- window bits shrink until they fit; nothing is built (not an error) when no width
 * fits, when n >= 2^16, when ZG_LAT_FULL_C = 0 or when the card has not that much free memory plus a reserve.
 * *bytes_built (may be NULL) = what is resident afterwards.  The tables belong to the base sets (shared by every prover on
 * them, freed with zg_bases_free); the throughput form never reads them.  Idempotent.  Proof bytes do not depend on it.
 * halo2 has no counterpart (ParamsKZG holds the bases only); the call site served is one synchronous proof,
 * /root/reference/src/wnn.rs:242-259. */
int zg_prover_enable_digit_tables(zg_prover *p, uint64_t max_bytes, uint64_t *bytes_built);

/* Stand-alone building blocks of the above. */
/* Evaluator::evaluate_h (halo2_proofs src/plonk/evaluation.rs) followed by the division by X^n - 1 of
 * vanishing::Argument::construct, for ONE circuit instance over this prover's resident proving key.  Inputs are the
 * coefficient forms upstream hands its evaluator -- advice_polys [n_advice][2^k], instance_polys [n_instance][2^k],
 * perm_z_polys [sets][2^k] (permutation::Committed product polys), lookup_z_polys [lookups][2^k] and permuted_polys
 * [2 * lookups][2^k] (a'_0, s'_0, a'_1, ...) -- and the challenges theta, beta, gamma, y; h_out receives h on
 * EvaluationDomain's extended coset, 2^ext_k values.  Host pointers (the arithmetic-level entry: ~60 MiB cross PCIe
 * at k = 14); slot 0 of the prover is used as scratch. */
int zg_prover_evaluate_h(zg_prover *p, const zg_fr *advice_polys, const zg_fr *instance_polys, const zg_fr *perm_z_polys,
                         const zg_fr *lookup_z_polys, const zg_fr *permuted_polys, const zg_fr *theta, const zg_fr *beta,
                         const zg_fr *gamma, const zg_fr *y, zg_fr *h_out);
/* z[0] = z0, z[i+1] = z[i] * num[i] / den[i], i + 1 < n  (the running product of lookup::prover::commit_product and
 * permutation::prover::commit; a zero denominator gives ratio 0, as halo2's BatchInvert leaves zeros alone).
 * Host pointers / device pointers. */
int zg_grand_product(zg_ctx *ctx, const zg_fr *num, const zg_fr *den, const zg_fr *z0, size_t n, zg_fr *z);
int zg_grand_product_dev(zg_ctx *ctx, const void *d_num, const void *d_den, const zg_fr *z0, size_t n,
                         void *d_z);
/* eval_polynomial: out[j] = polys[j](points[j]) for `count` (poly, point) pairs; polys at
 * d_polys + poly_index[j]*stride_elems; points are host scalars; out is a host array. */
int zg_eval_polys_dev(zg_ctx *ctx, const void *d_polys, size_t stride_elems, size_t n,
                      const uint32_t *poly_index, const zg_fr *points, size_t count, zg_fr *out);
/* kate_division: d_q[0..n-1) = a(X) / (X - z). */
int zg_kate_division_dev(zg_ctx *ctx, const void *d_a, size_t n, const zg_fr *z, void *d_q);
/* Keccak-256 (EvmTranscript's hash), host. */
void zg_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);

/* ---- witness of a batch of inputs on the device ------------------------------------------
 * Replaces, for a circuit whose layout does not depend on its input: the host pass upstream's create_proof makes
 * through Circuit::synthesize under its WitnessCollection (halo2_proofs v2023_04_20 src/plonk/prover.rs) -- for zero_g
 * WnnChip::predict, /root/reference/src/gadgets/wnn.rs:180-237, reached once per image from Wnn::proof,
 * /root/reference/src/wnn.rs:232-262.  The caller records that pass ONCE as a straight-line program over unsigned
 * 256-bit integers (one slot per operation; operands are earlier slots) and the library replays it per input:
 *
 *   op  0 CONST  consts[imm]        1 PIXEL  input byte imm     2 ADD a+b    3 SUB a-b     4 MUL a*b (low 256 bits)
 *       5 ADDI a+imm   6 RSUBI imm-a   7 MULI a*imm   8 SHRI a>>imm   9 SHLI a<<imm   10 ANDI a&imm   11 SHRV a>>b
 *       12 GTI a>imm   13 GEI a>=imm   14 EQI a==imm  (0 / 1)    15 DIVI a/imm    16 TABLE table[imm+a] (0 outside)
 *
 * Operations are given sorted by dependency level (level_start[l] .. level_start[l+1]); an operand must lie in an
 * earlier level (checked here, with every immediate: the program runs on the GPU unchecked).  cell_slot[c * 2^k + row]
 * names the slot advice column c shows in that row, or 0xFFFFFFFF for a cell left unassigned (zero); values are
 * reduced modulo r and written in the Montgomery form create_proof takes.  instance_slots: the public inputs. */
typedef struct zg_witness_plan zg_witness_plan;
typedef struct { uint64_t op, a, b, imm; } zg_witness_op;
int zg_witness_plan_create(zg_ctx* ctx, const zg_witness_op* ops, size_t n_ops, const uint32_t* level_start, size_t n_levels,
                           const uint64_t* consts /* [n_consts][4] little-endian words */, size_t n_consts,
                           const uint64_t* table, size_t n_table, const uint32_t* cell_slot /* [n_advice][2^k] */,
                           uint32_t n_advice, uint32_t k, const uint32_t* instance_slots, size_t n_instance,
                           size_t image_bytes, zg_witness_plan** out);
void zg_witness_plan_destroy(zg_witness_plan* plan);
size_t zg_witness_plan_image_bytes(const zg_witness_plan* plan);
size_t zg_witness_plan_instance_len(const zg_witness_plan* plan);
/* How the plan was laid out on the device (diagnostics; upstream has no counterpart -- its synthesis runs on the host):
 * out[0] bytes of LDS per workgroup (0: every operand in HBM, ZG_WITNESS_LDS = 0), out[1] / out[2] 8-byte / 32-byte LDS cells,
 * out[3] values that live in an LDS cell, out[4] consumed values left in HBM, out[5] levels whose barrier waits for HBM,
 * out[6] values whose interval bound exceeds 64 bits, out[7] levels, out[8] operations that run on 64-bit integers (operands
 * and result below 2^64 by their bounds), out[9] lanes of the workgroup. */
int zg_witness_plan_info(const zg_witness_plan* plan, uint64_t* out, size_t cap);
/* images: host, count * image_bytes; d_advice[i]: device, [n_advice][2^k] zg_fr (e.g. zg_prover_advice_slot);
 * instance_out: host, [count][n_instance].  At most 64 inputs per call.  Returns when the columns are written. */
int zg_witness_run_dev(zg_witness_plan* plan, const uint8_t* images, size_t count, void* const* d_advice, zg_fr* instance_out);
/* Wnn::proof for a batch (/root/reference/src/wnn.rs:232-262: image -> witness -> create_proof -> (proof bytes,
 * outputs)): the witness program into the prover's slots 0..count-1, then one lock-step batch of create_proofs with the
 * program's instance values as the single instance column.  outputs: host, [count][n_instance] (the class scores, as
 * field elements); everything else as zg_prover_prove_batch.  plan and prover must live on the same device. */
int zg_prover_prove_images(zg_prover* p, zg_witness_plan* plan, const uint8_t* images, size_t count, const uint8_t* rng_keys,
                           uint8_t* const* proofs, size_t proof_cap, size_t* proof_lens, zg_fr* outputs, int* statuses);

#ifdef __cplusplus
}
#endif
#endif /* ZG_HALO2_H */

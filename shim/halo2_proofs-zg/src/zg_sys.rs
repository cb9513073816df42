//! UNVERIFIED.  `extern "C"` view of include/zg_halo2.h (only what the shim calls).  Layouts: `zg_fr` / `zg_fq` are
//! four little-endian u64 Montgomery limbs = `bn256::Fr` / `Fq` in memory; `zg_g1_affine` = `G1Affine {x, y}`
//! ((0,0) = identity); `zg_g1` = `G1 {x, y, z}`.
#![allow(non_camel_case_types, dead_code)]
use halo2curves::bn256::{Fr, G1Affine, G1};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct zg_ctx { _p: [u8; 0] }
#[repr(C)] pub struct zg_bases { _p: [u8; 0] }
#[repr(C)] pub struct zg_prover { _p: [u8; 0] }
#[repr(C)] pub struct zg_witness_plan { _p: [u8; 0] }
/// one operation of a recorded witness program (include/zg_halo2.h: opcode table there)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct zg_witness_op { pub op: u64, pub a: u64, pub b: u64, pub imm: u64 }

pub const ZG_FIXED: u32 = 0;
pub const ZG_ADVICE: u32 = 1;
pub const ZG_INSTANCE: u32 = 2;
pub const ZG_MAX_FACTORS: usize = 8;
pub const ZG_MAX_LOOKUP_WIDTH: usize = 4;
pub const ZG_ERR_CONSTRAINT: c_int = -5;

#[repr(C)] #[derive(Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Debug)]
pub struct zg_query { pub kind: u32, pub column: u32, pub rotation: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct zg_monomial { pub coeff: Fr, pub n_factors: u32, pub factors: [u32; ZG_MAX_FACTORS] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct zg_poly { pub first: u32, pub count: u32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct zg_lookup { pub width: u32, pub inputs: [zg_poly; ZG_MAX_LOOKUP_WIDTH], pub tables: [zg_poly; ZG_MAX_LOOKUP_WIDTH] }
/// field order of `zg_circuit` in include/zg_halo2.h
#[repr(C)]
pub struct zg_circuit {
    pub k: u32, pub cs_degree: u32, pub blinding_factors: u32,
    pub n_fixed: u32, pub n_advice: u32, pub n_instance: u32,
    pub n_queries: u32, pub queries: *const zg_query,
    pub n_monomials: u32, pub monomials: *const zg_monomial,
    pub n_gates: u32, pub gates: *const zg_poly,
    pub n_lookups: u32, pub lookups: *const zg_lookup,
    pub n_perm_columns: u32, pub perm_columns: *const zg_query,
    pub n_advice_queries: u32, pub advice_queries: *const zg_query,
    pub n_fixed_queries: u32, pub fixed_queries: *const zg_query,
}

pub type zg_exchange_fn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, bytes: usize, recv: *mut c_void) -> c_int>;

extern "C" {
    pub fn zg_last_error() -> *const c_char;
    pub fn zg_ctx_create(device_id: c_int, out: *mut *mut zg_ctx) -> c_int;
    pub fn zg_ctx_destroy(ctx: *mut zg_ctx);
    pub fn zg_bases_register(ctx: *mut zg_ctx, bases: *const G1Affine, n: usize, window_bits: u32, out: *mut *mut zg_bases) -> c_int;
    pub fn zg_bases_free(b: *mut zg_bases);
    pub fn zg_msm(ctx: *mut zg_ctx, bases: *const zg_bases, scalars: *const Fr, n: usize, out: *mut G1) -> c_int;
    pub fn zg_ntt(ctx: *mut zg_ctx, a: *mut Fr, log_n: u32, omega: *const Fr) -> c_int;
    pub fn zg_intt(ctx: *mut zg_ctx, a: *mut Fr, log_n: u32, omega_inv: *const Fr, divisor: *const Fr) -> c_int;
    pub fn zg_coeff_to_extended(ctx: *mut zg_ctx, coeffs: *const Fr, k: u32, ext_k: u32, out: *mut Fr) -> c_int;
    pub fn zg_extended_to_coeff(ctx: *mut zg_ctx, evals: *mut Fr, k: u32, ext_k: u32, out_len: usize, out: *mut Fr) -> c_int;
    pub fn zg_grand_product(ctx: *mut zg_ctx, num: *const Fr, den: *const Fr, z0: *const Fr, n: usize, z: *mut Fr) -> c_int;
    pub fn zg_prover_create(ctx: *mut zg_ctx, cs: *const zg_circuit, fixed: *const Fr, sigma: *const Fr, g: *const G1Affine,
                            g_lagrange: *const G1Affine, vk_repr: *const Fr, out: *mut *mut zg_prover) -> c_int;
    pub fn zg_prover_create_shared(ctx: *mut zg_ctx, cs: *const zg_circuit, fixed: *const Fr, sigma: *const Fr, g: *const zg_bases,
                                   g_lagrange: *const zg_bases, vk_repr: *const Fr, out: *mut *mut zg_prover) -> c_int;
    pub fn zg_prover_fork(parent: *const zg_prover, ctx: *mut zg_ctx, out: *mut *mut zg_prover) -> c_int;
    pub fn zg_prover_destroy(p: *mut zg_prover);
    pub fn zg_prover_set_batch(p: *mut zg_prover, max_batch: usize) -> c_int;
    pub fn zg_prover_set_overlap(p: *mut zg_prover, enable: c_int) -> c_int;
    pub fn zg_prover_proof_size(p: *const zg_prover) -> usize;
    pub fn zg_prover_prove(p: *mut zg_prover, advice: *const Fr, instance: *const Fr, instance_len: usize, rng_key: *const u8,
                           proof: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn zg_prover_prove_batch(p: *mut zg_prover, count: usize, advice: *const *const Fr, instance: *const *const Fr,
                                 instance_len: usize, rng_keys: *const u8, proofs: *const *mut u8, cap: usize,
                                 lens: *mut usize, statuses: *mut c_int) -> c_int;
    pub fn zg_prover_evaluate_h(p: *mut zg_prover, advice_polys: *const Fr, instance_polys: *const Fr, perm_z_polys: *const Fr,
                                lookup_z_polys: *const Fr, permuted_polys: *const Fr, theta: *const Fr, beta: *const Fr,
                                gamma: *const Fr, y: *const Fr, h_out: *mut Fr) -> c_int;
    pub fn zg_prover_set_shard(p: *mut zg_prover, rank: u32, world: u32, first_point: usize, exchange: zg_exchange_fn,
                               user: *mut c_void) -> c_int;
    pub fn zg_bases_enable_bit_table(ctx: *mut zg_ctx, bases: *mut zg_bases, digit_width: u32) -> c_int;
    pub fn zg_prover_set_shard_rccl(p: *mut zg_prover, rank: u32, world: u32, first_point: usize, nccl_comm: *mut c_void) -> c_int;
    pub fn zg_xyzz_sum_ranks_dev(ctx: *mut zg_ctx, d_parts: *const c_void, world: usize, count: usize, d_out: *mut c_void) -> c_int;
    // witness of a batch of inputs on the device (optional; the recorder is the caller's: see README.md)
    pub fn zg_prover_advice_slot(p: *mut zg_prover, slot: usize) -> *mut c_void;
    pub fn zg_witness_plan_create(ctx: *mut zg_ctx, ops: *const zg_witness_op, n_ops: usize, level_start: *const u32, n_levels: usize,
                                  consts: *const u64, n_consts: usize, table: *const u64, n_table: usize, cell_slot: *const u32,
                                  n_advice: u32, k: u32, instance_slots: *const u32, n_instance: usize, image_bytes: usize,
                                  out: *mut *mut zg_witness_plan) -> c_int;
    pub fn zg_witness_plan_destroy(plan: *mut zg_witness_plan);
    pub fn zg_witness_run_dev(plan: *mut zg_witness_plan, images: *const u8, count: usize, d_advice: *const *mut c_void,
                              instance_out: *mut Fr) -> c_int;
    pub fn zg_prover_prove_images(p: *mut zg_prover, plan: *mut zg_witness_plan, images: *const u8, count: usize, rng_keys: *const u8,
                                  proofs: *const *mut u8, proof_cap: usize, proof_lens: *mut usize, outputs: *mut Fr,
                                  statuses: *mut c_int) -> c_int;
}

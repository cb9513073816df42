//! UNVERIFIED SOURCE -- never compiled (no Rust toolchain in the build image); written against halo2_proofs
//! v2023_04_20 from memory of the published crate.  See ../README.md.
//!
//! Routes `plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<_>, _, _, EvmTranscript<..>, _>` for ONE circuit
//! -- the instantiation zero_g uses (/root/reference/src/wnn.rs:242-259) -- to libzg_halo2.so:
//!
//!   * once per `ProvingKey`: `pk.vk.cs` is flattened into a `zg_circuit` (`flatten`: queries, every gate / lookup
//!     expression expanded into monomials by `expand`, permutation columns, advice / fixed query order) and handed to
//!     `zg_prover_create` with `pk.fixed_values`, `pk.permutation.permutations`, `params.g`, `params.g_lagrange` and
//!     `pk.vk.transcript_repr`; the prover (proving key resident in HBM) is cached by CONTENT (`Fingerprint`);
//!   * per proof: the witness is synthesised exactly as upstream's prover does (`WitnessCollection`, one phase), the
//!     `Assigned` values are batch-inverted, the advice columns go to `zg_prover_prove` with the instance values and 32
//!     bytes from the caller's RNG as blinding key, and the returned bytes -- the EvmTranscript stream -- are replayed
//!     into the caller's transcript object (`write_point` / `write_scalar` in proof order), so that
//!     `transcript.finalize()` in `Wnn::proof` returns them.
//!
//! Everything that does not fit returns `None` and the stock prover runs.
use std::any::TypeId;
use std::collections::{BTreeMap, HashMap};
use std::ffi::CStr;
use std::os::raw::c_int;
use std::sync::Mutex;

use ff::{Field, PrimeField};
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2curves::CurveAffine;
use once_cell::sync::Lazy;
use rand_core::RngCore;

use crate::circuit::Value;
use crate::plonk::{
    Advice, Any as AnyColumn, Assigned, Assignment, Challenge, Circuit, Column, ConstraintSystem, Error, Expression, Fixed,
    FloorPlanner, Instance, ProvingKey, Selector,
};
use crate::poly::commitment::{CommitmentScheme, Params};
use crate::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG};
use crate::poly::Rotation;
use crate::transcript::{EncodedChallenge, TranscriptWrite};
use crate::zg_sys::*;

fn check(status: c_int) -> Result<(), Error> {
    if status == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(zg_last_error()) }.to_string_lossy().into_owned();
    match status {
        ZG_ERR_CONSTRAINT => Err(Error::ConstraintSystemFailure), // a lookup input outside its table
        _ => panic!("zg_halo2 status {status}: {msg}"),            // upstream's arithmetic is infallible: nothing to map to
    }
}

// ------------------------------------------------------------------------------------------------ flattening
/// sorted query indices -> coefficient: a polynomial in expanded form
type Poly = BTreeMap<Vec<u32>, Fr>;

#[derive(Default)]
struct Queries {
    list: Vec<zg_query>,
    index: HashMap<(u32, u32, i32), u32>,
}
impl Queries {
    fn id(&mut self, kind: u32, column: usize, rotation: Rotation) -> u32 {
        let key = (kind, column as u32, rotation.0);
        if let Some(i) = self.index.get(&key) {
            return *i;
        }
        let i = self.list.len() as u32;
        self.list.push(zg_query { kind, column: column as u32, rotation: rotation.0 });
        self.index.insert(key, i);
        i
    }
}

/// `Expression<Fr>` -> sum of coeff * prod cell(query): exact over the field, so the value is the one halo2's
/// GraphEvaluator computes.  After keygen every `Selector` is a fixed-column query in `pk.vk.cs` already.
/// (harness/circuit.py `Expr` is the executable specification of this walk.)
fn expand(e: &Expression<Fr>, q: &mut Queries) -> Option<Poly> {
    fn one(ix: u32) -> Poly {
        [(vec![ix], Fr::one())].into_iter().collect()
    }
    Some(match e {
        Expression::Constant(c) => {
            if bool::from(c.is_zero()) { Poly::new() } else { [(vec![], *c)].into_iter().collect() }
        }
        Expression::Fixed(f) => one(q.id(ZG_FIXED, f.column_index(), f.rotation())),
        Expression::Advice(a) => one(q.id(ZG_ADVICE, a.column_index(), a.rotation())),
        Expression::Instance(i) => one(q.id(ZG_INSTANCE, i.column_index(), i.rotation())),
        Expression::Negated(a) => expand(a, q)?.into_iter().map(|(k, v)| (k, -v)).collect(),
        Expression::Scaled(a, s) => {
            let mut r: Poly = expand(a, q)?.into_iter().map(|(k, v)| (k, v * s)).collect();
            r.retain(|_, v| !bool::from(v.is_zero()));
            r
        }
        Expression::Sum(a, b) => {
            let mut r = expand(a, q)?;
            for (k, v) in expand(b, q)? {
                *r.entry(k).or_insert_with(Fr::zero) += v;
            }
            r.retain(|_, v| !bool::from(v.is_zero()));
            r
        }
        Expression::Product(a, b) => {
            let (x, y) = (expand(a, q)?, expand(b, q)?);
            let mut r = Poly::new();
            for (ka, va) in &x {
                for (kb, vb) in &y {
                    let mut k = [ka.as_slice(), kb.as_slice()].concat();
                    k.sort_unstable();
                    if k.len() > ZG_MAX_FACTORS {
                        return None; // a monomial of more than 8 cells: outside what the kernels are built for
                    }
                    *r.entry(k).or_insert_with(Fr::zero) += *va * vb;
                }
            }
            r.retain(|_, v| !bool::from(v.is_zero()));
            r
        }
        Expression::Selector(_) | Expression::Challenge(_) => return None, // not compressed / multi-phase: stock prover
    })
}

/// The arrays a `zg_circuit` points into; kept alive next to the prover handle.
struct Flat {
    queries: Vec<zg_query>,
    monomials: Vec<zg_monomial>,
    gates: Vec<zg_poly>,
    lookups: Vec<zg_lookup>,
    perm_columns: Vec<zg_query>,
    advice_queries: Vec<zg_query>,
    fixed_queries: Vec<zg_query>,
}

fn push_poly(p: Poly, monomials: &mut Vec<zg_monomial>) -> zg_poly {
    let first = monomials.len() as u32;
    for (cells, coeff) in p {
        let mut factors = [0u32; ZG_MAX_FACTORS];
        factors[..cells.len()].copy_from_slice(&cells);
        monomials.push(zg_monomial { coeff, n_factors: cells.len() as u32, factors });
    }
    zg_poly { first, count: monomials.len() as u32 - first }
}

fn column_query(c: &Column<AnyColumn>) -> zg_query {
    let kind = match c.column_type() {
        AnyColumn::Fixed => ZG_FIXED,
        AnyColumn::Advice(_) => ZG_ADVICE,
        AnyColumn::Instance => ZG_INSTANCE,
    };
    zg_query { kind, column: c.index() as u32, rotation: 0 }
}

fn flatten(cs: &ConstraintSystem<Fr>) -> Option<Flat> {
    if cs.num_challenges() != 0 || cs.phases().count() != 1 {
        return None;
    }
    let mut q = Queries::default();
    let mut monomials = Vec::new();
    let mut gates = Vec::new();
    for gate in cs.gates() {
        for poly in gate.polynomials() {
            gates.push(push_poly(expand(poly, &mut q)?, &mut monomials)); // gate polynomials in creation order
        }
    }
    let mut lookups = Vec::new();
    for l in cs.lookups() {
        let (ins, tabs) = (l.input_expressions(), l.table_expressions());
        if ins.len() != tabs.len() || ins.is_empty() || ins.len() > ZG_MAX_LOOKUP_WIDTH {
            return None;
        }
        let mut zl = zg_lookup { width: ins.len() as u32, inputs: [zg_poly::default(); 4], tables: [zg_poly::default(); 4] };
        for (i, (a, t)) in ins.iter().zip(tabs.iter()).enumerate() {
            zl.inputs[i] = push_poly(expand(a, &mut q)?, &mut monomials);
            zl.tables[i] = push_poly(expand(t, &mut q)?, &mut monomials);
        }
        lookups.push(zl);
    }
    let as_query = |kind: u32| move |(c, r): &(usize, Rotation)| zg_query { kind, column: *c as u32, rotation: r.0 };
    let advice_queries = cs.advice_queries().iter().map(|(c, r)| (c.index(), *r)).collect::<Vec<_>>();
    let fixed_queries = cs.fixed_queries().iter().map(|(c, r)| (c.index(), *r)).collect::<Vec<_>>();
    Some(Flat {
        queries: q.list,
        monomials,
        gates,
        lookups,
        perm_columns: cs.permutation().get_columns().iter().map(column_query).collect(),
        advice_queries: advice_queries.iter().map(as_query(ZG_ADVICE)).collect(),
        fixed_queries: fixed_queries.iter().map(as_query(ZG_FIXED)).collect(),
    })
}

// ------------------------------------------------------------------------------------------------ prover cache
struct Handle {
    ctx: *mut zg_ctx,
    prover: *mut zg_prover,
    proof_cap: usize,
    n_advice: usize,
    n: usize,
    _flat: Flat,
}
unsafe impl Send for Handle {} // every call into the library takes the context's lock

/// What a resident prover was built FROM, by content: the verifying key's transcript representation (a hash of the
/// whole constraint system and the fixed / permutation commitments), k, and the SRS through its second point
/// g[1] = [s]_1 (which fixes the toxic scalar and with it every base).  A key's ADDRESS is no identity: after a
/// `ProvingKey` is dropped another key or circuit can be allocated at the same address and would silently get the stale
/// GPU prover.
#[derive(Clone, Copy, PartialEq, Eq, Hash)]
struct Fingerprint {
    vk_repr: [u8; 32],
    k: u32,
    srs_g1: [u8; 64],
    batch: bool, // the lock-step form keeps a handle of its own: its slot count and scheduling never leak into lone proofs
}

fn fingerprint(params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>, batch: bool) -> Fingerprint {
    let mut vk_repr = [0u8; 32];
    vk_repr.copy_from_slice(pk.get_vk().transcript_repr.to_repr().as_ref());
    let mut srs_g1 = [0u8; 64];
    if let Some(p) = params.get_g().get(1) {
        let c = p.coordinates().unwrap();
        srs_g1[..32].copy_from_slice(c.x().to_repr().as_ref());
        srs_g1[32..].copy_from_slice(c.y().to_repr().as_ref());
    }
    Fingerprint { vk_repr, k: params.k(), srs_g1, batch }
}

/// one resident prover per (proving key, SRS) content and form
static PROVERS: Lazy<Mutex<HashMap<Fingerprint, Handle>>> = Lazy::new(|| Mutex::new(HashMap::new()));

fn device() -> c_int {
    std::env::var("ZG_HALO2_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}

fn build_handle(params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>) -> Option<Handle> {
    let cs = pk.vk.cs();
    let flat = flatten(cs)?;
    let n = params.n() as usize;
    let c = zg_circuit {
        k: params.k(),
        cs_degree: cs.degree() as u32,
        blinding_factors: cs.blinding_factors() as u32,
        n_fixed: cs.num_fixed_columns() as u32,
        n_advice: cs.num_advice_columns() as u32,
        n_instance: cs.num_instance_columns() as u32,
        n_queries: flat.queries.len() as u32, queries: flat.queries.as_ptr(),
        n_monomials: flat.monomials.len() as u32, monomials: flat.monomials.as_ptr(),
        n_gates: flat.gates.len() as u32, gates: flat.gates.as_ptr(),
        n_lookups: flat.lookups.len() as u32, lookups: flat.lookups.as_ptr(),
        n_perm_columns: flat.perm_columns.len() as u32, perm_columns: flat.perm_columns.as_ptr(),
        n_advice_queries: flat.advice_queries.len() as u32, advice_queries: flat.advice_queries.as_ptr(),
        n_fixed_queries: flat.fixed_queries.len() as u32, fixed_queries: flat.fixed_queries.as_ptr(),
    };
    // [n_fixed][n] and [n_perm_columns][n] Lagrange values, contiguous (Polynomial<Fr, LagrangeCoeff> derefs to [Fr])
    let fixed: Vec<Fr> = pk.fixed_values.iter().flat_map(|p| p.iter().copied()).collect();
    let sigma: Vec<Fr> = pk.permutation.permutations.iter().flat_map(|p| p.iter().copied()).collect();
    let vk_repr: Fr = pk.vk.transcript_repr;
    let mut ctx = std::ptr::null_mut();
    let mut prover = std::ptr::null_mut();
    unsafe {
        check(zg_ctx_create(device(), &mut ctx)).ok()?;
        check(zg_prover_create(ctx, &c, fixed.as_ptr(), sigma.as_ptr(), params.get_g().as_ptr(), params.g_lagrange.as_ptr(),
                               &vk_repr, &mut prover)).ok()?;
        Some(Handle { ctx, prover, proof_cap: zg_prover_proof_size(prover), n_advice: cs.num_advice_columns(), n, _flat: flat })
    }
}

// ------------------------------------------------------------------------------------------------ witness
/// upstream prover.rs `WitnessCollection`, single phase: records advice assignments, checks instance queries.
/// Generic over the field so that `try_create_proof` can call it with `Scheme::Scalar` (which IS Fr there).
struct WitnessCollection<'a, F: Field> {
    k: u32,
    advice: Vec<Vec<Assigned<F>>>,
    instances: &'a [&'a [F]],
    usable_rows: std::ops::RangeTo<usize>,
}

impl<'a, F: Field> Assignment<F> for WitnessCollection<'a, F> {
    fn enter_region<NR, N>(&mut self, _: N) where NR: Into<String>, N: FnOnce() -> NR {}
    fn exit_region(&mut self) {}
    fn enable_selector<A, AR>(&mut self, _: A, _: &Selector, _: usize) -> Result<(), Error>
    where A: FnOnce() -> AR, AR: Into<String> { Ok(()) }
    fn annotate_column<A, AR>(&mut self, _: A, _: Column<AnyColumn>) where A: FnOnce() -> AR, AR: Into<String> {}
    fn query_instance(&self, column: Column<Instance>, row: usize) -> Result<Value<F>, Error> {
        if !self.usable_rows.contains(&row) {
            return Err(Error::not_enough_rows_available(self.k));
        }
        self.instances.get(column.index()).and_then(|c| c.get(row)).map(|v| Value::known(*v)).ok_or(Error::BoundsFailure)
    }
    fn assign_advice<V, VR, A, AR>(&mut self, _: A, column: Column<Advice>, row: usize, to: V) -> Result<(), Error>
    where V: FnOnce() -> Value<VR>, VR: Into<Assigned<F>>, A: FnOnce() -> AR, AR: Into<String> {
        if !self.usable_rows.contains(&row) {
            return Err(Error::not_enough_rows_available(self.k));
        }
        *self.advice.get_mut(column.index()).and_then(|v| v.get_mut(row)).ok_or(Error::BoundsFailure)? =
            to().into_field().assign()?;
        Ok(())
    }
    fn assign_fixed<V, VR, A, AR>(&mut self, _: A, _: Column<Fixed>, _: usize, _: V) -> Result<(), Error>
    where V: FnOnce() -> Value<VR>, VR: Into<Assigned<F>>, A: FnOnce() -> AR, AR: Into<String> { Ok(()) }
    fn copy(&mut self, _: Column<AnyColumn>, _: usize, _: Column<AnyColumn>, _: usize) -> Result<(), Error> { Ok(()) }
    fn fill_from_row(&mut self, _: Column<Fixed>, _: usize, _: Value<Assigned<F>>) -> Result<(), Error> { Ok(()) }
    fn get_challenge(&self, _: Challenge) -> Value<F> { Value::unknown() }
    fn push_namespace<NR, N>(&mut self, _: N) where NR: Into<String>, N: FnOnce() -> NR {}
    fn pop_namespace(&mut self, _: Option<String>) {}
}

/// advice columns of one circuit instance, [n_advice][n] contiguous, as create_proof has them before blinding
fn synthesize<F: Field, ConcreteCircuit: Circuit<F>>(cs: &ConstraintSystem<F>, k: u32, circuit: &ConcreteCircuit,
                                                     instances: &[&[F]]) -> Result<Vec<F>, Error> {
    let n = 1usize << k;
    let unusable = cs.blinding_factors() + 1;
    #[cfg(feature = "circuit-params")]
    let config = { let mut meta = ConstraintSystem::default(); ConcreteCircuit::configure_with_params(&mut meta, circuit.params()) };
    #[cfg(not(feature = "circuit-params"))]
    let config = { let mut meta = ConstraintSystem::default(); ConcreteCircuit::configure(&mut meta) };
    let mut witness = WitnessCollection {
        k,
        advice: vec![vec![Assigned::Zero; n]; cs.num_advice_columns()],
        instances,
        usable_rows: ..n - unusable,
    };
    ConcreteCircuit::FloorPlanner::synthesize(&mut witness, circuit, config, cs.constants().clone())?;
    // what batch_invert_assigned does to Polynomial<Assigned<F>, LagrangeCoeff>, on plain vectors
    let mut out = Vec::with_capacity(cs.num_advice_columns() * n);
    for column in witness.advice {
        let mut denoms: Vec<F> = column.iter().map(|a| a.denominator().unwrap_or(F::ONE)).collect();
        // (Montgomery's trick; zero denominators stay zero as in upstream's BatchInvert)
        let mut acc = F::ONE;
        let mut prefix = Vec::with_capacity(n);
        for d in &denoms { prefix.push(acc); if !bool::from(d.is_zero()) { acc *= d; } }
        let mut inv = acc.invert().unwrap_or(F::ZERO);
        for (d, p) in denoms.iter_mut().zip(prefix.iter()).rev() {
            if bool::from(d.is_zero()) { continue; }
            let t = inv * p; inv *= *d; *d = t;
        }
        out.extend(column.iter().zip(denoms.iter()).map(|(a, dinv)| a.numerator() * dinv));
    }
    Ok(out)
}

// ------------------------------------------------------------------------------------------------ entry point
fn from_be32<F: PrimeField>(b: &[u8]) -> Option<F> {
    let mut repr = F::Repr::default();
    let le = repr.as_mut();
    if le.len() != 32 { return None; }
    for i in 0..32 { le[i] = b[31 - i]; }
    Option::from(F::from_repr(repr))
}

/// Proof bytes in EvmTranscript layout (points as x || y, scalars, 32-byte big-endian each) ->
/// the caller's transcript object, in proof order.  Generic: C is G1Affine whenever this runs.
fn replay<C: CurveAffine, E: EncodedChallenge<C>, T: TranscriptWrite<C, E>>(cs: &ConstraintSystem<C::Scalar>, proof: &[u8],
                                                                            transcript: &mut T) -> std::io::Result<()> {
    let bad = |what: &'static str| std::io::Error::new(std::io::ErrorKind::InvalidData, what);
    let nl = cs.lookups().len();
    let chunk = cs.degree() - 2;
    let sets = (cs.permutation().get_columns().len() + chunk - 1) / chunk;
    let points_before_evals = cs.num_advice_columns() + 2 * nl + sets + nl + 1 + (cs.degree() - 1);
    let scalars = cs.advice_queries().len() + cs.fixed_queries().len() + 1 + cs.permutation().get_columns().len()
        + if sets > 0 { 3 * sets - 1 } else { 0 } + 5 * nl;
    if proof.len() < 64 * points_before_evals + 32 * scalars || (proof.len() - 64 * points_before_evals - 32 * scalars) % 64 != 0 {
        return Err(bad("zg proof: unexpected length"));
    }
    let mut at = 0usize;
    let mut point = |t: &mut T, at: &mut usize| -> std::io::Result<()> {
        let x = from_be32::<C::Base>(&proof[*at..*at + 32]).ok_or_else(|| bad("zg proof: bad coordinate"))?;
        let y = from_be32::<C::Base>(&proof[*at + 32..*at + 64]).ok_or_else(|| bad("zg proof: bad coordinate"))?;
        *at += 64;
        let p = Option::<C>::from(C::from_xy(x, y)).ok_or_else(|| bad("zg proof: point not on the curve"))?;
        t.write_point(p)
    };
    for _ in 0..points_before_evals { point(transcript, &mut at)?; }
    for _ in 0..scalars {
        let s = from_be32::<C::Scalar>(&proof[at..at + 32]).ok_or_else(|| bad("zg proof: bad scalar"))?;
        at += 32;
        transcript.write_scalar(s)?;
    }
    while at < proof.len() { point(transcript, &mut at)?; } // the GWC witness commitments, one per opening point
    Ok(())
}

/// First statement of `plonk::create_proof` in the fork (README step 5).  `None` = not zero_g's instantiation: the stock
/// prover runs.  The fork's `create_proof` also has the prover type `P` in scope and should guard the call with
/// `TypeId::of::<P>() == TypeId::of::<ProverGWC<'params, Bn256>>()`; SHPLONK proofs have another opening layout.
pub fn try_create_proof<'params, Scheme, E, R, T, ConcreteCircuit>(
    params: &'params Scheme::ParamsProver,
    pk: &ProvingKey<Scheme::Curve>,
    circuits: &[ConcreteCircuit],
    instances: &[&[&[Scheme::Scalar]]],
    rng: &mut R,
    transcript: &mut T,
) -> Option<Result<(), Error>>
where
    Scheme: CommitmentScheme + 'static,
    E: EncodedChallenge<Scheme::Curve>,
    R: RngCore,
    T: TranscriptWrite<Scheme::Curve, E>,
    ConcreteCircuit: Circuit<Scheme::Scalar>,
{
    if std::env::var_os("ZG_HALO2_DISABLE").is_some() || circuits.len() != 1 || instances.len() != 1 {
        return None;
    }
    if TypeId::of::<Scheme>() != TypeId::of::<KZGCommitmentScheme<Bn256>>() {
        return None;
    }
    // The transcript must be snark-verifier's Keccak EvmTranscript: its byte stream is what the library returns.  That
    // type cannot be named from halo2_proofs (dependency cycle), so the application opts in: zero_g sets
    // ZG_HALO2_TRANSCRIPT=evm (its only transcript for create_proof, /root/reference/src/wnn.rs:249).
    if std::env::var("ZG_HALO2_TRANSCRIPT").as_deref() != Ok("evm") {
        return None;
    }
    // From here on Scheme IS KZGCommitmentScheme<Bn256>: Scheme::Curve = G1Affine, Scheme::Scalar = Fr and
    // Scheme::ParamsProver = ParamsKZG<Bn256>.  The thin-pointer casts below are identities.
    let params_kzg: &ParamsKZG<Bn256> = unsafe { &*(params as *const Scheme::ParamsProver as *const ParamsKZG<Bn256>) };
    let pk_g1: &ProvingKey<G1Affine> = unsafe { &*(pk as *const ProvingKey<Scheme::Curve> as *const ProvingKey<G1Affine>) };

    let key = fingerprint(params_kzg, pk_g1, false);
    let mut cache = PROVERS.lock().unwrap();
    if !cache.contains_key(&key) {
        let h = build_handle(params_kzg, pk_g1)?;
        // The lone-proof digit tables (78 GB at k = 14, ~0.8 s to build) are the APPLICATION's decision since round 4: zero_g
        // opts in with ZG_HALO2_DIGIT_TABLES=<max bytes> (0 = the library's cap); without it a lone proof keeps the bucket
        // form (2.9 instead of 2.2 ms at k = 14) and the card keeps its memory.
        if let Some(v) = std::env::var_os("ZG_HALO2_DIGIT_TABLES") {
            let max_bytes: u64 = v.to_string_lossy().parse().unwrap_or(0);
            let mut built: u64 = 0;
            if unsafe { zg_prover_enable_digit_tables(h.prover, max_bytes, &mut built) } != 0 {
                return None; // (the tables are an optimisation: a failure to build them is not a failure to prove -- but say so)
            }
        }
        cache.insert(key, h);
    }
    let h = cache.get(&key)?;
    let cs = pk.get_vk().cs(); // ConstraintSystem<Scheme::Scalar>
    Some((|| {
        let instances = instances[0];
        if instances.len() != cs.num_instance_columns() {
            return Err(Error::InvalidInstances);
        }
        let inst_len = instances.iter().map(|c| c.len()).max().unwrap_or(0);
        if inst_len > h.n - (cs.blinding_factors() + 1) {
            return Err(Error::InstanceTooLarge);
        }
        // [n_instance][inst_len], zero-padded like upstream's instance polynomials
        let mut inst = vec![Scheme::Scalar::ZERO; instances.len() * inst_len];
        for (c, col) in instances.iter().enumerate() {
            inst[c * inst_len..c * inst_len + col.len()].copy_from_slice(col);
        }
        let advice: Vec<Scheme::Scalar> = synthesize(cs, params_kzg.k(), &circuits[0], instances)?;
        debug_assert_eq!(advice.len(), h.n_advice * h.n);
        let mut key32 = [0u8; 32];
        rng.fill_bytes(&mut key32); // upstream draws every blinding scalar from this RNG; here it keys their generator
        let mut buf = vec![0u8; h.proof_cap];
        let mut len = 0usize;
        check(unsafe {
            zg_prover_prove(h.prover, advice.as_ptr() as *const Fr, inst.as_ptr() as *const Fr, inst_len, key32.as_ptr(),
                            buf.as_mut_ptr(), buf.len(), &mut len)
        })?;
        replay::<Scheme::Curve, E, T>(cs, &buf[..len], transcript).map_err(Error::from)
    })())
}

// ------------------------------------------------------------------------------------------------ batch API
/// Lock-step batches for callers that prove many images (`zero_g` over MNIST test images, BASELINE configs[4]): one
/// resident prover, `witnesses.len()` proofs per launch sequence.  Returns the proofs' bytes (EvmTranscript streams).
pub fn create_proofs_zg<ConcreteCircuit: Circuit<Fr>, R: RngCore>(
    params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>, circuits: &[ConcreteCircuit], instances: &[&[&[Fr]]], rng: &mut R,
) -> Result<Vec<Vec<u8>>, Error> {
    assert_eq!(circuits.len(), instances.len());
    // (a handle of its own: the slot count and the throughput scheduling set here must not leak into the lone-proof
    //  path of try_create_proof, which keeps its latency form)
    let key = fingerprint(params, pk, true);
    let mut cache = PROVERS.lock().unwrap();
    if !cache.contains_key(&key) {
        let h = build_handle(params, pk).expect("circuit outside the backend's scope");
        check(unsafe { zg_prover_set_overlap(h.prover, 0) })?; // throughput form
        cache.insert(key, h);
    }
    let h = cache.get(&key).unwrap();
    let count = circuits.len();
    if unsafe { zg_prover_batch(h.prover) } < count {
        check(unsafe { zg_prover_set_batch(h.prover, count) })?;
    }
    let inst_len = instances[0].iter().map(|c| c.len()).max().unwrap_or(0);
    let mut adv = Vec::with_capacity(count);
    let mut inst = Vec::with_capacity(count);
    for (c, i) in circuits.iter().zip(instances.iter()) {
        adv.push(synthesize(pk.get_vk().cs(), params.k(), c, i)?); // (rayon: witness synthesis of the batch in parallel)
        let mut flat = vec![Fr::zero(); i.len() * inst_len];
        for (col, v) in i.iter().enumerate() { flat[col * inst_len..col * inst_len + v.len()].copy_from_slice(v); }
        inst.push(flat);
    }
    let adv_ptrs: Vec<*const Fr> = adv.iter().map(|a| a.as_ptr()).collect();
    let inst_ptrs: Vec<*const Fr> = inst.iter().map(|a| a.as_ptr()).collect();
    let mut keys = vec![0u8; 32 * count];
    rng.fill_bytes(&mut keys);
    let mut bufs = vec![vec![0u8; h.proof_cap]; count];
    let out_ptrs: Vec<*mut u8> = bufs.iter_mut().map(|b| b.as_mut_ptr()).collect();
    let mut lens = vec![0usize; count];
    let mut sts = vec![0 as c_int; count];
    check(unsafe { zg_prover_prove_batch(h.prover, count, adv_ptrs.as_ptr(), inst_ptrs.as_ptr(), inst_len, keys.as_ptr(),
                                         out_ptrs.as_ptr(), h.proof_cap, lens.as_mut_ptr(), sts.as_mut_ptr()) })?;
    for (b, l) in bufs.iter_mut().zip(lens) { b.truncate(l); }
    Ok(bufs)
}

impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { zg_prover_destroy(self.prover); zg_ctx_destroy(self.ctx); }
    }
}

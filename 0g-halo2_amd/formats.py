"""On-disk artefacts either side of the proving path, as zero_g's CLI writes and reads them
(/root/reference/src/io.rs:137-207), so that a backend process can consume what `zero_g generate-srs`
produced and hand back what `zero_g verify` reads:

  write_srs / read_srs            io.rs:139-146  `ParamsKZG::<Bn256>::{write, read}`
  write/read_circuit_params       io.rs:149-156  serde_json of `WnnCircuitParams`
  write_pk / read_pk              io.rs:159-169  `ProvingKey::<G1Affine>::{write, read}` with `SerdeFormat::RawBytes`
  ProofWithOutput.write / .read   io.rs:179-207  serde_json of `{proof: Vec<u8>, output: Vec<Fr>}`

Byte layouts follow halo2_proofs v2023_04_20 / halo2curves 0.3.3 as published -- those crates are not
in the reference checkout and no file written by the real CLI exists here, so the layouts are
[UPSTREAM-MEMORY] and only round-trip-tested:
  * ParamsKZG::write = write_custom(.., SerdeFormat::RawBytes): k as u32 LE; n G1Affine of g; n G1Affine
    of g_lagrange; g2; s_g2.  RawBytes point = coordinates as raw Montgomery limbs, little-endian
    (G1Affine 64 B = x, y; G2Affine 128 B = x.c0, x.c1, y.c0, y.c1) -- i.e. exactly the in-memory
    arrays the C ABI takes (`zg_g1_affine`), which is why this reader is a header parse + two views.
  * `Fr` inside ProofWithOutput: the reference derives serde on the struct (io.rs:179-183) with halo2curves'
    `derive_serde` feature.  Two encodings are plausible for 0.3.3 and neither can be checked here: the derive sitting
    on the newtype `Fr([u64; 4])` (JSON: an array of the four Montgomery limbs -- the ABI's zg_fr) or a hand-written
    impl over `to_repr` (JSON: "0x" + 64 hex digits of the canonical little-endian bytes).  read() accepts both,
    write(form=...) produces either; interoperability with `zero_g verify` is NOT claimed until one real file is seen.
  * ProvingKey::write (halo2_proofs src/plonk.rs at v2023_04_20), field by field; every count is a u32 BIG-endian,
    every scalar 32 B of raw Montgomery limbs (RawBytes), every point 64 B (x, y):
        VerifyingKey::write:
            k                                   u32 BE
            number of fixed commitments         u32 BE
            fixed_commitments                   that many G1Affine
            permutation VerifyingKey            one G1Affine per permutation column (no count: the reader knows
                                                cs.permutation.columns.len() from the circuit)
            selectors                           cs.num_selectors bit-vectors of 2^k bits, packed 8 rows per byte,
                                                least significant bit = lowest row (no count either)
        l0, l_last, l_active_row                three Polynomial::write: length u32 BE (= 2^ext_k), then the values
                                                (extended-domain Lagrange form)
        fixed_values, fixed_polys, fixed_cosets three write_polynomial_slice: count u32 BE, then that many
                                                Polynomial::write (2^k Lagrange values / 2^k coefficients / 2^ext_k
                                                coset values)
        permutation ProvingKey                  three write_polynomial_slice: permutations (2^k Lagrange values of
                                                the sigma polynomials), polys (coefficients), cosets (2^ext_k)
    read_pk hands back every field; zg_prover_create needs only `fixed_values` and `permutations` (it derives the
    coefficient and coset forms itself, on the GPU) plus vk.transcript_repr, which is a Blake2b hash over the Debug
    print of the pinned verifying key and therefore has to come from the Rust side.
"""
from __future__ import annotations

import json
import struct
from dataclasses import asdict, dataclass

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # BN254 scalar field
MONT = (1 << 256) % R


@dataclass
class WnnCircuitParams:
    """gadgets/wnn.rs `WnnCircuitParams` as filled by Wnn::get_circuit_params (/root/reference/src/wnn.rs:171-181):
    the file `zero_g` keeps next to its keys (io.rs:149-156)."""
    p: int
    l: int
    n_hashes: int
    bits_per_hash: int
    bits_per_filter: int
    n_classes: int

G1_BYTES, G2_BYTES = 64, 128


def write_srs(path: str, k: int, g: np.ndarray, g_lagrange: np.ndarray, g2: np.ndarray, s_g2: np.ndarray):
    """g, g_lagrange: uint64[n, 8] Montgomery limbs (x, y); g2, s_g2: uint64[16] (x.c0, x.c1, y.c0, y.c1)."""
    n = 1 << k
    g = np.ascontiguousarray(g, dtype="<u8").reshape(n, 8)
    g_lagrange = np.ascontiguousarray(g_lagrange, dtype="<u8").reshape(n, 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", k))
        f.write(g.tobytes())
        f.write(g_lagrange.tobytes())
        f.write(np.ascontiguousarray(g2, dtype="<u8").reshape(16).tobytes())
        f.write(np.ascontiguousarray(s_g2, dtype="<u8").reshape(16).tobytes())


def read_srs(path: str):
    """-> (k, g, g_lagrange, g2, s_g2); the point arrays are memory-mapped views (288 GB of HBM on the
    other side: the file is streamed to the device, never duplicated on the host)."""
    with open(path, "rb") as f:
        (k,) = struct.unpack("<I", f.read(4))
    if not 1 <= k <= 28:
        raise ValueError(f"{path}: implausible k = {k}")
    n = 1 << k
    expect = 4 + 2 * n * G1_BYTES + 2 * G2_BYTES
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    if mm.shape[0] != expect:
        raise ValueError(f"{path}: {mm.shape[0]} bytes, expected {expect} for k = {k}")
    g = mm[4:4 + n * G1_BYTES].view("<u8").reshape(n, 8)
    gl = mm[4 + n * G1_BYTES:4 + 2 * n * G1_BYTES].view("<u8").reshape(n, 8)
    tail = np.array(mm[4 + 2 * n * G1_BYTES:]).view("<u8")
    return k, g, gl, tail[:16].copy(), tail[16:].copy()


def write_circuit_params(path: str, params: WnnCircuitParams):
    with open(path, "w") as f:
        json.dump(asdict(params), f, separators=(",", ":"))


def read_circuit_params(path: str) -> WnnCircuitParams:
    with open(path) as f:
        d = json.load(f)
    return WnnCircuitParams(**{k: int(d[k]) for k in ("p", "l", "n_hashes", "bits_per_hash", "bits_per_filter", "n_classes")})


def fr_to_repr_hex(x: int) -> str:
    return "0x" + (x % R).to_bytes(32, "little").hex()


def fr_from_repr_hex(s: str) -> int:
    raw = bytes.fromhex(s[2:] if s.startswith("0x") else s)
    if len(raw) != 32:
        raise ValueError("Fr repr must be 32 bytes")
    x = int.from_bytes(raw, "little")
    if x >= R:
        raise ValueError("non-canonical Fr")
    return x


class ProofWithOutput:
    """io.rs:179-207: the circuit's public output (class scores) next to the proof bytes."""

    def __init__(self, proof: bytes, output: list):
        self.proof, self.output = bytes(proof), [int(v) % R for v in output]

    def write(self, path: str, form: str = "limbs"):
        """form = "limbs": every output as the array of its four Montgomery u64 limbs (serde derive on the newtype);
        "hex": as "0x" + the canonical little-endian bytes in hex (module docstring: which one halo2curves 0.3.3 emits is
        unconfirmed)."""
        if form == "limbs":
            m = (1 << 64) - 1
            out = [[(v * MONT % R >> (64 * j)) & m for j in range(4)] for v in self.output]
        elif form == "hex":
            out = [fr_to_repr_hex(v) for v in self.output]
        else:
            raise ValueError(form)
        with open(path, "w") as f:
            json.dump({"proof": list(self.proof), "output": out}, f, separators=(",", ":"))

    @staticmethod
    def read(path: str) -> "ProofWithOutput":
        with open(path) as f:
            d = json.load(f)
        out = []
        for v in d["output"]:
            if isinstance(v, str):
                out.append(fr_from_repr_hex(v))
            else:
                if len(v) != 4 or any(not 0 <= int(x) < 1 << 64 for x in v):
                    raise ValueError("Fr limbs must be four u64")
                x = sum(int(l) << (64 * j) for j, l in enumerate(v))
                if x >= R:
                    raise ValueError("non-canonical Fr limbs")
                out.append(x * pow(MONT, -1, R) % R)
        return ProofWithOutput(bytes(d["proof"]), out)

    def output_mont(self) -> np.ndarray:
        """uint64[1, len, 4]: the instance column in the ABI's Montgomery limb format."""
        m = (1 << 64) - 1
        out = np.zeros((1, len(self.output), 4), np.uint64)
        for i, v in enumerate(self.output):
            x = v * MONT % R
            out[0, i] = [(x >> (64 * j)) & m for j in range(4)]
        return out


class ProvingKeyFile:
    """What ProvingKey::write(RawBytes) holds (module docstring): arrays are uint64 limb arrays, points [.., 8],
    scalars [.., 4]; selectors is bool[num_selectors, 2^k]."""

    FIELDS = ("fixed_commitments", "permutation_commitments", "selectors", "l0", "l_last", "l_active_row", "fixed_values",
              "fixed_polys", "fixed_cosets", "permutations", "permutation_polys", "permutation_cosets")

    def __init__(self, k: int, **kw):
        self.k = k
        for f in self.FIELDS:
            setattr(self, f, kw[f])


def _be32(x: int) -> bytes:
    return struct.pack(">I", x)


def _poly_bytes(v: np.ndarray) -> bytes:
    v = np.ascontiguousarray(v, dtype="<u8").reshape(-1, 4)
    return _be32(v.shape[0]) + v.tobytes()


def _slice_bytes(polys: np.ndarray) -> bytes:
    polys = np.ascontiguousarray(polys, dtype="<u8")
    return _be32(polys.shape[0]) + b"".join(_poly_bytes(p) for p in polys)


def write_pk(path: str, pk: ProvingKeyFile):
    n = 1 << pk.k
    sel = np.asarray(pk.selectors, dtype=bool).reshape(-1, n)
    with open(path, "wb") as f:
        f.write(_be32(pk.k))
        fc = np.ascontiguousarray(pk.fixed_commitments, dtype="<u8").reshape(-1, 8)
        f.write(_be32(fc.shape[0]))
        f.write(fc.tobytes())
        f.write(np.ascontiguousarray(pk.permutation_commitments, dtype="<u8").reshape(-1, 8).tobytes())
        for row in sel:
            f.write(np.packbits(row, bitorder="little").tobytes())
        for poly in (pk.l0, pk.l_last, pk.l_active_row):
            f.write(_poly_bytes(poly))
        for sl in (pk.fixed_values, pk.fixed_polys, pk.fixed_cosets, pk.permutations, pk.permutation_polys, pk.permutation_cosets):
            f.write(_slice_bytes(sl))


def read_pk(path: str, num_selectors: int, num_permutation_columns: int) -> ProvingKeyFile:
    """num_selectors = cs.num_selectors BEFORE selector compression, num_permutation_columns =
    cs.permutation.columns.len(): upstream's reader re-runs Circuit::configure to learn them (io.rs:163-168 passes the
    circuit parameters for that), this one is told.  Polynomial families come back as memory-mapped views."""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    pos = 0

    def take(nbytes):
        nonlocal pos
        if pos + nbytes > mm.shape[0]:
            raise ValueError(f"{path}: truncated at byte {pos} (+{nbytes})")
        out = mm[pos:pos + nbytes]
        pos += nbytes
        return out

    def be32():
        return struct.unpack(">I", bytes(take(4)))[0]

    def poly(expect=None):
        ln = be32()
        if expect is not None and ln != expect:
            raise ValueError(f"{path}: polynomial of {ln} values where {expect} were expected")
        return take(ln * 32).view("<u8").reshape(ln, 4)

    def poly_slice(expect=None):
        cnt = be32()
        if cnt > 1 << 16:
            raise ValueError(f"{path}: implausible polynomial count {cnt}")
        polys = [poly(expect) for _ in range(cnt)]
        return np.stack(polys) if polys else np.zeros((0, expect or 0, 4), "<u8")

    k = be32()
    if not 1 <= k <= 28:
        raise ValueError(f"{path}: implausible k = {k}")
    n = 1 << k
    nfc = be32()
    if nfc > 1 << 16:
        raise ValueError(f"{path}: implausible fixed-commitment count {nfc}")
    fixed_commitments = take(nfc * G1_BYTES).view("<u8").reshape(nfc, 8)
    perm_commitments = take(num_permutation_columns * G1_BYTES).view("<u8").reshape(num_permutation_columns, 8)
    selectors = np.zeros((num_selectors, n), dtype=bool)
    for i in range(num_selectors):
        selectors[i] = np.unpackbits(np.array(take((n + 7) // 8)), bitorder="little")[:n].astype(bool)
    l0 = poly()
    en = l0.shape[0]
    if en < n or en & (en - 1):
        raise ValueError(f"{path}: l0 has {en} values for k = {k}")
    l_last, l_active = poly(en), poly(en)
    fixed_values, fixed_polys, fixed_cosets = poly_slice(n), poly_slice(n), poly_slice(en)
    perms, perm_polys, perm_cosets = poly_slice(n), poly_slice(n), poly_slice(en)
    if pos != mm.shape[0]:
        raise ValueError(f"{path}: {mm.shape[0] - pos} bytes left over")
    if fixed_values.shape[0] != nfc or perms.shape[0] != num_permutation_columns:
        raise ValueError(f"{path}: {fixed_values.shape[0]} fixed columns / {perms.shape[0]} permutation columns do not match the "
                         f"verifying key ({nfc} / {num_permutation_columns})")
    return ProvingKeyFile(k, fixed_commitments=fixed_commitments, permutation_commitments=perm_commitments, selectors=selectors,
                          l0=l0, l_last=l_last, l_active_row=l_active, fixed_values=fixed_values, fixed_polys=fixed_polys,
                          fixed_cosets=fixed_cosets, permutations=perms, permutation_polys=perm_polys,
                          permutation_cosets=perm_cosets)

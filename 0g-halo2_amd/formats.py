"""On-disk artefacts either side of the proving path, as zero_g's CLI writes and reads them
(/root/reference/src/io.rs:137-207), so that a backend process can consume what `zero_g generate-srs`
produced and hand back what `zero_g verify` reads:

  write_srs / read_srs            io.rs:139-146  `ParamsKZG::<Bn256>::{write, read}`
  write/read_circuit_params       io.rs:149-156  serde_json of `WnnCircuitParams`
  ProofWithOutput.write / .read   io.rs:179-207  serde_json of `{proof: Vec<u8>, output: Vec<Fr>}`

Byte layouts follow halo2_proofs v2023_04_20 / halo2curves 0.3.3 as published -- those crates are not
in the reference checkout and no file written by the real CLI exists here, so the layouts are
[UPSTREAM-MEMORY] and only round-trip-tested:
  * ParamsKZG::write = write_custom(.., SerdeFormat::RawBytes): k as u32 LE; n G1Affine of g; n G1Affine
    of g_lagrange; g2; s_g2.  RawBytes point = coordinates as raw Montgomery limbs, little-endian
    (G1Affine 64 B = x, y; G2Affine 128 B = x.c0, x.c1, y.c0, y.c1) -- i.e. exactly the in-memory
    arrays the C ABI takes (`zg_g1_affine`), which is why this reader is a header parse + two views.
  * halo2curves' `Fr` with its serde feature serialises as the canonical 32-byte little-endian
    representation, hex-encoded ("0x" + 64 digits) in JSON.
ProvingKey / VerifyingKey files (`pk.write(writer, RawBytes)`, io.rs:159-176) are not parsed: their
layout interleaves the verifying key, selector bit-vectors and five polynomial families in an order
that cannot be checked here; the ABI takes their content as flat arrays instead.
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

    def write(self, path: str):
        with open(path, "w") as f:
            json.dump({"proof": list(self.proof), "output": [fr_to_repr_hex(v) for v in self.output]}, f,
                      separators=(",", ":"))

    @staticmethod
    def read(path: str) -> "ProofWithOutput":
        with open(path) as f:
            d = json.load(f)
        return ProofWithOutput(bytes(d["proof"]), [fr_from_repr_hex(v) for v in d["output"]])

    def output_mont(self) -> np.ndarray:
        """uint64[1, len, 4]: the instance column in the ABI's Montgomery limb format."""
        m = (1 << 64) - 1
        out = np.zeros((1, len(self.output), 4), np.uint64)
        for i, v in enumerate(self.output):
            x = v * MONT % R
            out[0, i] = [(x >> (64 * j)) & m for j in range(4)]
        return out

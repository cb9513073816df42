/* Struct layouts of include/zg_halo2.h as the C compiler sees them (C99, LP64) -- what Rust's #[repr(C)] structs in
 * shim/halo2_proofs-zg/src/zg_sys.rs and the ctypes Structures in harness/circuit.py / 0g-halo2_amd/zg_halo2.py must
 * reproduce.  Compile-time: the numbers below are asserted (a header change that moves a field stops the build).
 * Run-time: prints the same numbers as JSON; tests/test_abi.py compares them with ctypes.sizeof / field offsets and
 * with the layout it computes from the Rust struct declarations.
 *     gcc -std=c99 -pedantic -Wall -Werror -I include tests/abi/layout.c -o tests/abi/layout && tests/abi/layout */
#include <stddef.h>
#include <stdio.h>

#include "zg_halo2.h"

#define STATIC_ASSERT(name, cond) typedef char static_assert_##name[(cond) ? 1 : -1]

STATIC_ASSERT(fr_size, sizeof(zg_fr) == 32);
STATIC_ASSERT(fq_size, sizeof(zg_fq) == 32);
STATIC_ASSERT(affine_size, sizeof(zg_g1_affine) == 64 && offsetof(zg_g1_affine, y) == 32);
STATIC_ASSERT(g1_size, sizeof(zg_g1) == 96 && offsetof(zg_g1, z) == 64);
STATIC_ASSERT(query, sizeof(zg_query) == 12 && offsetof(zg_query, column) == 4 && offsetof(zg_query, rotation) == 8);
STATIC_ASSERT(monomial, sizeof(zg_monomial) == 72 && offsetof(zg_monomial, n_factors) == 32 && offsetof(zg_monomial, factors) == 36);
STATIC_ASSERT(poly, sizeof(zg_poly) == 8 && offsetof(zg_poly, count) == 4);
STATIC_ASSERT(lookup, sizeof(zg_lookup) == 68 && offsetof(zg_lookup, inputs) == 4 && offsetof(zg_lookup, tables) == 36);
STATIC_ASSERT(circuit, sizeof(zg_circuit) == 136 && offsetof(zg_circuit, n_queries) == 24 && offsetof(zg_circuit, queries) == 32 &&
                           offsetof(zg_circuit, n_monomials) == 40 && offsetof(zg_circuit, monomials) == 48 &&
                           offsetof(zg_circuit, n_gates) == 56 && offsetof(zg_circuit, gates) == 64 &&
                           offsetof(zg_circuit, n_lookups) == 72 && offsetof(zg_circuit, lookups) == 80 &&
                           offsetof(zg_circuit, n_perm_columns) == 88 && offsetof(zg_circuit, perm_columns) == 96 &&
                           offsetof(zg_circuit, n_advice_queries) == 104 && offsetof(zg_circuit, advice_queries) == 112 &&
                           offsetof(zg_circuit, n_fixed_queries) == 120 && offsetof(zg_circuit, fixed_queries) == 128);
STATIC_ASSERT(witness_op, sizeof(zg_witness_op) == 32 && offsetof(zg_witness_op, imm) == 24);
STATIC_ASSERT(kernel_stat, sizeof(zg_kernel_stat) == 80 && offsetof(zg_kernel_stat, unit_bytes) == 72 && offsetof(zg_kernel_stat, launches) == 48 &&
                               offsetof(zg_kernel_stat, total_ms) == 56 && offsetof(zg_kernel_stat, algo_bytes) == 64);
STATIC_ASSERT(status_codes, ZG_OK == 0 && ZG_ERR_INVALID_ARG == -1 && ZG_ERR_NO_DEVICE == -2 && ZG_ERR_HIP == -3 &&
                                ZG_ERR_UNSUPPORTED == -4 && ZG_ERR_CONSTRAINT == -5 && ZG_ERR_OOM == -6);
STATIC_ASSERT(enums, ZG_FIXED == 0 && ZG_ADVICE == 1 && ZG_INSTANCE == 2 && ZG_MAX_FACTORS == 8 && ZG_MAX_LOOKUP_WIDTH == 4);

#define FIELD(type, field) printf("%s\"%s\": %lu", first++ ? ", " : "", #field, (unsigned long)offsetof(type, field))
#define BEGIN(type) printf("%s\"%s\": {\"size\": %lu, \"fields\": {", structs++ ? ",\n " : " ", #type, (unsigned long)sizeof(type)); first = 0
#define END() printf("}}")

int main(void) {
    int first = 0, structs = 0;
    printf("{\n");
    BEGIN(zg_query); FIELD(zg_query, kind); FIELD(zg_query, column); FIELD(zg_query, rotation); END();
    BEGIN(zg_monomial); FIELD(zg_monomial, coeff); FIELD(zg_monomial, n_factors); FIELD(zg_monomial, factors); END();
    BEGIN(zg_poly); FIELD(zg_poly, first); FIELD(zg_poly, count); END();
    BEGIN(zg_lookup); FIELD(zg_lookup, width); FIELD(zg_lookup, inputs); FIELD(zg_lookup, tables); END();
    BEGIN(zg_circuit);
    FIELD(zg_circuit, k); FIELD(zg_circuit, cs_degree); FIELD(zg_circuit, blinding_factors); FIELD(zg_circuit, n_fixed);
    FIELD(zg_circuit, n_advice); FIELD(zg_circuit, n_instance); FIELD(zg_circuit, n_queries); FIELD(zg_circuit, queries);
    FIELD(zg_circuit, n_monomials); FIELD(zg_circuit, monomials); FIELD(zg_circuit, n_gates); FIELD(zg_circuit, gates);
    FIELD(zg_circuit, n_lookups); FIELD(zg_circuit, lookups); FIELD(zg_circuit, n_perm_columns); FIELD(zg_circuit, perm_columns);
    FIELD(zg_circuit, n_advice_queries); FIELD(zg_circuit, advice_queries); FIELD(zg_circuit, n_fixed_queries);
    FIELD(zg_circuit, fixed_queries);
    END();
    BEGIN(zg_witness_op); FIELD(zg_witness_op, op); FIELD(zg_witness_op, a); FIELD(zg_witness_op, b); FIELD(zg_witness_op, imm); END();
    BEGIN(zg_kernel_stat); FIELD(zg_kernel_stat, name); FIELD(zg_kernel_stat, launches); FIELD(zg_kernel_stat, total_ms);
    FIELD(zg_kernel_stat, algo_bytes); FIELD(zg_kernel_stat, unit_bytes); END();
    printf("\n}\n");
    return 0;
}

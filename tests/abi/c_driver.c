/* The drop-in boundary driven from plain C -- the call shape a Rust `extern "C"` binding makes
 * (/root/reference/src/wnn.rs:242-259 -> shim/halo2_proofs-zg/src/zg.rs), with no Python in the process:
 *   ParamsKZG::new(k)            -> zg_params_new
 *   best_multiexp(col, g_lagrange) -> zg_bases_register + zg_msm
 *   create_proof                  -> zg_prover_create + zg_prover_prove, then a lock-step batch of two
 * on the statement of c_driver_data.h.  Prints one line per result ("msm <hex>", "proof <hex>", "batch0 <hex>",
 * "batch1 <hex>"); tests/test_gpu_abi_c.py compares them with the oracle's bytes.  C99, no extensions:
 *     gcc -std=c99 -pedantic -Wall -Werror -I include tests/abi/c_driver.c -L 0g-halo2_amd -lzg_halo2 -o tests/abi/c_driver */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zg_halo2.h"
#include "c_driver_data.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int st_ = (call);                                                             \
        if (st_ != ZG_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, zg_last_error());           \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static void hex(const char *label, const void *data, size_t len) {
    const unsigned char *p = (const unsigned char *)data;
    size_t i;
    printf("%s ", label);
    for (i = 0; i < len; i++) printf("%02x", p[i]);
    printf("\n");
}

int main(void) {
    const zg_circuit circuit = DRV_CIRCUIT;
    zg_ctx *ctx = NULL;
    zg_bases *gl_bases = NULL;
    zg_prover *prover = NULL;
    zg_g1_affine *g = malloc(sizeof(zg_g1_affine) * DRV_N), *gl = malloc(sizeof(zg_g1_affine) * DRV_N);
    zg_g1 sum;
    size_t cap, len = 0, lens[2] = {0, 0};
    uint8_t *proof, *pair[2];
    const zg_fr *adv2[2], *inst2[2];
    int sts[2] = {0, 0};

    if (!g || !gl) return 2;
    printf("version %s\n", zg_version());
    CHECK(zg_ctx_create(0, &ctx));
    CHECK(zg_params_new(ctx, DRV_K, &drv_s, g, gl));
    CHECK(zg_bases_register(ctx, gl, DRV_N, 0, &gl_bases));
    CHECK(zg_msm(ctx, gl_bases, drv_advice, DRV_N, &sum)); /* commit_lagrange of advice column 0 */
    hex("msm", &sum, sizeof(sum));

    CHECK(zg_prover_create(ctx, &circuit, drv_fixed, drv_sigma, g, gl, &drv_vk_repr, &prover));
    cap = zg_prover_proof_size(prover);
    proof = malloc(cap);
    pair[0] = malloc(cap);
    pair[1] = malloc(cap);
    if (!proof || !pair[0] || !pair[1]) return 2;
    CHECK(zg_prover_prove(prover, drv_advice, drv_instance, DRV_INSTANCE_LEN, drv_keys, proof, cap, &len));
    hex("proof", proof, len);

    CHECK(zg_prover_set_batch(prover, 2));
    CHECK(zg_prover_set_overlap(prover, 0)); /* throughput form */
    adv2[0] = adv2[1] = drv_advice;
    inst2[0] = inst2[1] = drv_instance;
    CHECK(zg_prover_prove_batch(prover, 2, adv2, inst2, DRV_INSTANCE_LEN, drv_keys + 32, pair, cap, lens, sts));
    if (sts[0] != ZG_OK || sts[1] != ZG_OK) return 3;
    hex("batch0", pair[0], lens[0]);
    hex("batch1", pair[1], lens[1]);

    zg_prover_destroy(prover);
    zg_bases_free(gl_bases);
    zg_ctx_destroy(ctx);
    free(proof); free(pair[0]); free(pair[1]); free(g); free(gl);
    return 0;
}

/* Plain-C consumer of include/mina_verify.h -- what a cgo / Rust `extern "C"` binding sees.  Built with gcc (no HIP
 * headers) and linked against libminaverify.so by tests/test_c_abi.py.  Exit code 0 = every check passed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mina_verify.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL %s (%s)\n", msg, mina_last_error()); return 1; } } while (0)

int main(void) {
    mina_ctx *ctx = NULL;
    CHECK(mina_ctx_create(0, &ctx) == MINA_OK && ctx, "ctx_create");
    CHECK(mina_srs_depth(ctx, MINA_CURVE_VESTA) == 0, "no SRS yet");
    const uint32_t k = 9, n = 1u << k;
    CHECK(mina_srs_create(ctx, MINA_CURVE_VESTA, n) == MINA_OK, "srs_create");
    CHECK(mina_srs_depth(ctx, MINA_CURVE_VESTA) == n, "srs_depth");

    /* errors are return codes, never crashes */
    uint8_t pt[64], pt2[64];
    CHECK(mina_msm_srs(ctx, MINA_CURVE_PALLAS, 4, (const uint8_t *)"", pt) == MINA_ERR_STATE, "SRS not loaded -> MINA_ERR_STATE");
    CHECK(mina_msm_srs(ctx, 7, 4, pt, pt) == MINA_ERR_ARG, "bad curve -> MINA_ERR_ARG");
    CHECK(mina_srs_load(ctx, MINA_CURVE_VESTA, (const uint8_t *)"garbage", 7) == MINA_ERR_FORMAT, "bad SRS blob -> MINA_ERR_FORMAT");
    CHECK(mina_srs_depth(ctx, MINA_CURVE_VESTA) == n, "failed load leaves the old SRS in place");

    /* linearity of the fixed-base MSM: MSM(e_3) == g[3], MSM(2 e_3) == g[3] + g[3] via the variable-base entry */
    uint8_t *sc = calloc(n, 32);
    sc[3 * 32] = 1;
    CHECK(mina_msm_srs(ctx, MINA_CURVE_VESTA, n, sc, pt) == MINA_OK, "msm_srs");
    uint8_t g3[64];
    CHECK(mina_srs_get_g(ctx, MINA_CURVE_VESTA, 3, 1, g3) == MINA_OK, "srs_get_g");
    CHECK(memcmp(pt, g3, 64) == 0, "MSM(e_3) == g[3]");
    sc[3 * 32] = 2;
    CHECK(mina_msm_srs(ctx, MINA_CURVE_VESTA, n, sc, pt) == MINA_OK, "msm_srs 2");
    uint8_t two_pts[128], ones[64] = {0};
    memcpy(two_pts, g3, 64); memcpy(two_pts + 64, g3, 64);
    ones[0] = 1; ones[32] = 1;
    CHECK(mina_msm(ctx, MINA_CURVE_VESTA, 2, two_pts, ones, pt2) == MINA_OK, "msm");
    CHECK(memcmp(pt, pt2, 64) == 0, "2*g[3] fixed-base == g[3]+g[3] variable-base");

    /* accumulator check: mint sg with the library, accept; perturb a prechallenge, reject */
    uint8_t pre[9 * 16], chals[9 * 32], *coef = malloc((size_t)n * 32), sg[64], verdict = 9;
    for (unsigned i = 0; i < sizeof pre; ++i) pre[i] = (uint8_t)(i * 37 + 11);
    CHECK(mina_challenge_to_field(ctx, MINA_FIELD_FP, k, pre, chals) == MINA_OK, "challenge_to_field");
    CHECK(mina_b_poly_coefficients(ctx, MINA_FIELD_FP, k, chals, coef) == MINA_OK, "b_poly_coefficients");
    CHECK(mina_msm_srs(ctx, MINA_CURVE_VESTA, n, coef, sg) == MINA_OK, "commit");
    CHECK(mina_accumulator_check_batch(ctx, MINA_CURVE_VESTA, k, 1, pre, sg, NULL, &verdict) == MINA_OK && verdict == 1, "accept");
    pre[5] ^= 1;
    CHECK(mina_accumulator_check_batch(ctx, MINA_CURVE_VESTA, k, 1, pre, sg, NULL, &verdict) == MINA_OK && verdict == 0, "reject");

    /* wire format parser: host-only */
    uint8_t pub[1057] = {0};
    mina_state_pub_inputs out;
    pub[0] = 1; pub[1] = 42;
    CHECK(mina_parse_state_pub_inputs(pub, sizeof pub, &out) == MINA_OK && out.is_state_proof_from_devnet == 1 && out.bridge_tip_state_hash[0] == 42, "parse pub inputs");
    CHECK(mina_parse_state_pub_inputs(pub, 1056, &out) == MINA_ERR_FORMAT, "wrong length -> MINA_ERR_FORMAT");

    free(sc); free(coef);
    mina_ctx_destroy(ctx);
    printf("c_abi_smoke ok\n");
    return 0;
}

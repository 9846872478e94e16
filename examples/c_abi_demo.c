/* Plain-C user of the drop-in boundary (include/stardis_hip.h): no Python, no torch, no HIP headers.
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lstardis_amd/lib -lstardis_hip -Wl,-rpath,$PWD/stardis_amd/lib -lm -o c_abi_demo
 * Builds a small deterministic problem, calls the two host-buffer entry points that replace the reference's numba kernels
 * (calc_alan_entries, opacities_solvers/base.py:487-592; the raytrace loop, radiation_field_solvers/base.py:271-346) and
 * prints every number it got back with 17 significant digits; tests/test_gpu_c_abi.py rebuilds the same inputs in Python
 * and checks the output against the CPU oracle.  It then runs the whole fused synthesis twice — sdx_synthesize_f64 on one
 * context, and sdx_synthesize_sharded_f64 on a group of every visible GPU (one RCCL all-gather of the emergent flux inside the
 * library) — and reports whether the two agree bit for bit. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stardis_hip.h"

static uint64_t state = 0x9E3779B97F4A7C15ull;
static double uniform(void) /* xorshift64*, the test reproduces it */
{
    state ^= state >> 12;
    state ^= state << 25;
    state ^= state >> 27;
    return (double)((state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}

int main(void)
{
    enum { ND = 5, NNU = 400, NL = 7, NTH = 3 };
    static double nus[NNU], line_nus[NL], dw[NL * ND], gam[NL * ND], al[NL * ND], out[ND * NNU], temps[ND], dist[(ND - 1) * NTH],
        wts[NTH], F[ND * NNU];
    for (int i = 0; i < NNU; ++i) nus[i] = 4.57e14 - 1.0e9 * i; /* descending grid */
    for (int l = 0; l < NL; ++l) line_nus[l] = nus[NNU - 1] + (nus[0] - nus[NNU - 1]) * (l + 0.5) / NL; /* ascending */
    for (int k = 0; k < NL * ND; ++k) {
        dw[k] = 2.0e9 * (1.0 + uniform());
        gam[k] = 1.0e8 * (1.0 + 9.0 * uniform());
        al[k] = pow(10.0, -3.0 + 4.0 * uniform());
    }
    for (int d = 0; d < ND; ++d) temps[d] = 9000.0 - 1000.0 * d;
    const double thetas[NTH] = {0.2, 0.8, 1.3};
    for (int g = 0; g < ND - 1; ++g)
        for (int t = 0; t < NTH; ++t) dist[g * NTH + t] = (1.0e7 * (1 + g)) / cos(thetas[t]); /* :302-305 */
    for (int t = 0; t < NTH; ++t) wts[t] = 0.3 + 0.1 * t;

    if (sdx_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", sdx_last_error_string());
        return 3;
    }
    sdx_ctx* ctx = sdx_create(0, NULL);
    if (!ctx) {
        fprintf(stderr, "sdx_create: %s\n", sdx_last_error_string());
        return 2;
    }
    int64_t evals = 0;
    int rc = sdx_line_opacity_f64(ctx, ND, NNU, nus, NL, line_nus, dw, gam, ND, al, out, &evals);
    if (rc) {
        fprintf(stderr, "sdx_line_opacity_f64: %d %s\n", rc, sdx_last_error_string());
        return 1;
    }
    for (int k = 0; k < ND * NNU; ++k) {
        out[k] += 1.0e-9; /* a grey continuum so that no opacity is zero */
        F[k] = 0.0;
    }
    rc = sdx_raytrace_f64(ctx, ND, NNU, NTH, nus, temps, dist, wts, out, F, NULL);
    if (rc) {
        fprintf(stderr, "sdx_raytrace_f64: %d %s\n", rc, sdx_last_error_string());
        return 1;
    }
    /* error path: an ascending grid must be refused with -1 */
    double bad[3] = {1.0, 2.0, 3.0}, o3[ND * 3];
    const int rc_bad = sdx_line_opacity_f64(ctx, ND, 3, bad, NL, line_nus, dw, gam, ND, al, o3, NULL);
    /* the fused synthesis on host buffers (electron scattering as the only continuum source), on one context and on a group of
     * all visible GPUs: the sharded call returns the same F_nu, and the gathered emergent flux is its last row */
    static double ne[ND], F1[ND * NNU], F2[ND * NNU], flux[NNU];
    for (int d = 0; d < ND; ++d) ne[d] = 1.0e17 / (1 + d); /* optical depths per gap of order 0.1-1: well inside the reference's exp branch (:36-45) */
    sdx_continuum cont;
    memset(&cont, 0, sizeof cont);
    cont.electron_density = ne;
    cont.temperature = temps;
    rc = sdx_synthesize_f64(ctx, ND, NNU, nus, NL, line_nus, dw, gam, ND, al, &cont, NTH, temps, dist, wts, NULL, NULL, F1, NULL);
    if (rc) {
        fprintf(stderr, "sdx_synthesize_f64: %d %s\n", rc, sdx_last_error_string());
        return 1;
    }
    const int n_gpus = sdx_device_count();
    sdx_group* grp = sdx_group_create(n_gpus, NULL);
    if (!grp) {
        fprintf(stderr, "sdx_group_create: %d %s\n", sdx_last_error_code(), sdx_last_error_string());
        return 1;
    }
    rc = sdx_synthesize_sharded_f64(grp, ND, NNU, nus, NL, line_nus, dw, gam, ND, al, &cont, NTH, temps, dist, wts, NULL, NULL, NULL, F2, flux, NULL);
    if (rc) {
        fprintf(stderr, "sdx_synthesize_sharded_f64: %d %s\n", rc, sdx_last_error_string());
        return 1;
    }
    int ranks = 0, rccl = 0;
    int64_t bytes = 0;
    sdx_group_last_gather(grp, &ranks, &bytes, &rccl);
    const int same = memcmp(F1, F2, sizeof F1) == 0 && memcmp(flux, F2 + (ND - 1) * NNU, sizeof flux) == 0;
    sdx_group_destroy(grp);
    printf("version %s\nevaluations %lld\nbad_grid_rc %d\nsharded ranks %d bytes_per_rank %lld rccl %d identical %d\n", sdx_version(), (long long)evals,
           rc_bad, ranks, (long long)bytes, rccl, same);
    for (int k = 0; k < ND * NNU; ++k) printf("%.17g %.17g %.17g\n", out[k], F[k], F1[k]);
    sdx_destroy(ctx);
    return 0;
}

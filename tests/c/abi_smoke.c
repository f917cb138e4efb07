/* C-ABI smoke test in plain C: proves include/mixlab_gpu.h is a C header and the library is usable
 * without any C++/Python on the caller's side.  Builds a 2-oscillator -> Mixer graph, runs 4 ticks,
 * reads Master back, checks a few structural facts, exercises the error channel.
 * Build:  gcc -std=c11 -I include tests/c/abi_smoke.c -L mixlab_amd -lmixlab_gpu -Wl,-rpath,$PWD/mixlab_amd -o abi_smoke -lm
 * Exit code 0 = ok, 2 = no GPU (reported, not a failure of the ABI), 1 = failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mixlab_gpu.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mx_last_error()); return 1; } } while (0)

int main(void) {
    if (mx_abi_version() != MX_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    if (mx_device_count() <= 0) { fprintf(stderr, "no GPU: %s\n", mx_last_error()); return 2; }

    mx_oscillator_params saw = {220.0, MX_WAVE_SAW, 0}, tri = {880.0, MX_WAVE_TRIANGLE, 0};
    mx_mixer_channel_params ch[2] = {{0.0, 1.0, 0, {0}}, {-6.0, 0.5, 1, {0}}};
    mx_node nodes[3] = {
        {MX_KIND_OSCILLATOR, sizeof saw, &saw}, {MX_KIND_OSCILLATOR, sizeof tri, &tri}, {MX_KIND_MIXER, sizeof ch, ch}};
    mx_edge edges[2] = {{0, 1, 2, 0}, {1, 1, 2, 1}};   /* stereo outs -> mixer inputs */
    mx_graph_opts opts; memset(&opts, 0, sizeof opts); opts.device = -1; opts.max_ticks_per_run = 4;
    mx_graph* g = NULL;
    CHECK(mx_graph_build(nodes, 3, edges, 2, &opts, &g));
    size_t spt = 0;
    CHECK(mx_graph_samples_per_tick(g, &spt));
    if (spt != 735) { fprintf(stderr, "SPT %zu != 735\n", spt); return 1; }
    CHECK(mx_graph_run_ticks(g, 0, 4));
    float* master = (float*)malloc(sizeof(float) * 2 * spt * 4);
    float* cue = (float*)malloc(sizeof(float) * 2 * spt * 4);
    CHECK(mx_graph_read_output(g, 2, 0, master, 4));
    CHECK(mx_graph_read_output(g, 2, 1, cue, 4));
    /* saw(0) = 0, triangle(0) = -1: master[0] = 0*1 + (-1 * 0.5 * 10^(-6/20)) ; cue[0] = -1 (channel 2 only) */
    double want = -0.5 * pow(10.0, -6.0 / 20.0);
    if (fabs(master[0] - want) > 1e-6 || master[0] != master[1] || cue[0] != -1.0f) {
        fprintf(stderr, "unexpected samples: master[0]=%g (want %g) cue[0]=%g\n", master[0], want, cue[0]); return 1;
    }
    /* type mismatch is an error code, not a crash (the reference refuses the connection, workspace.rs:97-114) */
    mx_eq_three_params eq = {0, 0, 0};
    mx_node bad_nodes[2] = {{MX_KIND_OSCILLATOR, sizeof saw, &saw}, {MX_KIND_EQ_THREE, sizeof eq, &eq}};
    mx_edge bad_edge = {0, 1, 1, 0};   /* Stereo -> Mono */
    mx_graph* g2 = NULL;
    int rc = mx_graph_build(bad_nodes, 2, &bad_edge, 1, &opts, &g2);
    if (rc != MX_ERR_TYPE || g2 != NULL || strlen(mx_last_error()) == 0) { fprintf(stderr, "type mismatch not reported (rc=%d)\n", rc); return 1; }
    /* per-module path: one EqThree, 100 samples in one call */
    mx_module* m = NULL;
    CHECK(mx_module_create(MX_KIND_EQ_THREE, &eq, sizeof eq, &m));
    float in[100], out[100];
    for (int i = 0; i < 100; i++) in[i] = (float)sin(i * 0.1);
    mx_input mi = {MX_MONO, in, 100, NULL};
    mx_output mo = {MX_MONO, out, 100, NULL, 0};
    size_t ind_len = 0;
    CHECK(mx_module_run_tick(m, 0, &mi, 1, &mo, 1, NULL, &ind_len));
    mx_module_destroy(m);
    mx_graph_destroy(g);
    free(master); free(cue);
    printf("abi_smoke ok: master[0]=%.6f eq_out[99]=%.6f\n", master ? want : 0.0, out[99]);
    return 0;
}

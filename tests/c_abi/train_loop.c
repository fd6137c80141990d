/* A host written in plain C against include/pvae.h: no Python, no torch -- only the HIP runtime
 * for device memory.  It is what a compiled caller of the drop-in boundary looks like:
 * query the layout, allocate the arenas, bind a demonstration set, run world-model steps and then
 * joint steps with pvae_train_step, read the losses back; then serve one observation through the launches and through the
 * call-persistent rollout server and compare the actions.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ tests/c_abi/train_loop.c -Iinclude -I/opt/rocm/include \
 *       -L/opt/rocm/lib -lamdhip64 -ldl -lm -o train_loop
 *   ./train_loop physicsvae_amd/libpvae_gfx950.so
 *
 * Exit code 0 and a last line "ok ..." when the losses are finite, the world-model loss fell during
 * the world phase and the reconstruction loss fell during the joint phase.  (Numerical parity is the
 * job of the Python tests; this program checks that the boundary is usable as a C ABI.) */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "pvae.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LOAD(name) name##_t name = (name##_t)dlsym(h, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define CHECK(call) do { int r_ = (call); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, r_, pvae_last_error()); return 3; } } while (0)

typedef int (*pvae_abi_version_t)(void);
typedef const char* (*pvae_last_error_t)(void);
typedef int (*pvae_num_layers_t)(const pvae_config*);
typedef int (*pvae_layer_t)(const pvae_config*, int, pvae_layer_info*);
typedef int64_t (*pvae_arena_floats_t)(const pvae_config*);
typedef size_t (*pvae_workspace_bytes_t)(const pvae_config*);
typedef int (*pvae_create_t)(const pvae_config*, pvae_ctx**);
typedef void (*pvae_destroy_t)(pvae_ctx*);
typedef int (*pvae_bind_arenas_t)(pvae_ctx*, float*, float*, float*, float*);
typedef int (*pvae_bind_workspace_t)(pvae_ctx*, void*, size_t);
typedef int (*pvae_bind_dataset_t)(pvae_ctx*, const float*, const float*, const int32_t*, int64_t, int64_t);
typedef int (*pvae_train_step_t)(pvae_ctx*, int, int64_t, int32_t, const pvae_step_params*, const float*, float*, void*);
typedef int (*pvae_infer_t)(pvae_ctx*, const float*, int32_t, const float*, int, uint64_t, uint64_t, float*, float*, float*, void*);
typedef int (*pvae_rollout_server_start_t)(pvae_ctx*, double, double, int);
typedef int (*pvae_rollout_server_infer_t)(pvae_ctx*, const float*, int, uint64_t, uint64_t, int, float*, float*, float*, double);
typedef int (*pvae_rollout_server_selfbench_t)(pvae_ctx*, const float*, int, int32_t, double*);
typedef int (*pvae_rollout_server_stop_t)(pvae_ctx*);

static uint32_t lcg_state = 12345u;
static float uniform(void) {                     /* (-1, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return ((lcg_state >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "physicsvae_amd/libpvae_gfx950.so";
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return 2; }
    LOAD(pvae_abi_version) LOAD(pvae_last_error) LOAD(pvae_num_layers) LOAD(pvae_layer) LOAD(pvae_arena_floats)
    LOAD(pvae_workspace_bytes) LOAD(pvae_create) LOAD(pvae_destroy) LOAD(pvae_bind_arenas) LOAD(pvae_bind_workspace)
    LOAD(pvae_bind_dataset) LOAD(pvae_train_step) LOAD(pvae_infer)
    LOAD(pvae_rollout_server_start) LOAD(pvae_rollout_server_infer) LOAD(pvae_rollout_server_selfbench) LOAD(pvae_rollout_server_stop)
    if (pvae_abi_version() != PVAE_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", pvae_abi_version(), PVAE_ABI_VERSION); return 2; }

    pvae_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.dim_body = 23; cfg.dim_action = 7; cfg.latent = 8;
    cfg.te_width = 64; cfg.te_depth = 2; cfg.md_width = 96; cfg.md_depth = 2; cfg.wm_width = 128; cfg.wm_depth = 2;
    cfg.max_batch = 64; cfg.lookahead = 1;
    const int Db = cfg.dim_body, Da = cfg.dim_action, B = 64;
    const int64_t nf = pvae_arena_floats(&cfg);
    const size_t wsb = pvae_workspace_bytes(&cfg);

    /* parameters: each weight row scaled like the trainer's normc initialiser (unit row norm for
     * hidden layers, 0.01 for output layers, zero bias); pad entries stay zero */
    float* hp = (float*)calloc((size_t)nf, sizeof(float));
    const int nl = pvae_num_layers(&cfg);
    for (int i = 0; i < nl; ++i) {
        pvae_layer_info li;
        CHECK(pvae_layer(&cfg, i, &li));
        int last = (i + 1 == nl);
        if (!last) { pvae_layer_info nx; CHECK(pvae_layer(&cfg, i + 1, &nx)); last = nx.net != li.net; }
        for (int r = 0; r < li.n_out; ++r) {
            double ss = 0;
            float* row = hp + li.w_offset + (int64_t)r * li.ld;
            for (int c = 0; c < li.n_in; ++c) { row[c] = uniform(); ss += (double)row[c] * row[c]; }
            const float s = (float)((last ? 0.01 : 1.0) / sqrt(ss));
            for (int c = 0; c < li.n_in; ++c) row[c] *= s;
        }
    }
    /* demonstrations: 4 episodes x 200 steps of a noisy linear system whose action is a fixed function
     * of the state (so both the world model and the encoder/decoder have something to learn);
     * windows = every row but the last of each episode */
    const int E = 4, T = 200;
    const int64_t n_rows = (int64_t)E * T, n_win = (int64_t)E * (T - 1);
    float* hs = (float*)malloc((size_t)n_rows * Db * sizeof(float));
    float* ha = (float*)malloc((size_t)n_rows * Da * sizeof(float));
    int32_t* hw = (int32_t*)malloc((size_t)n_win * sizeof(int32_t));
    for (int e = 0, w = 0; e < E; ++e)
        for (int t = 0; t < T; ++t) {
            const int64_t r = (int64_t)e * T + t;
            for (int c = 0; c < Db; ++c)
                hs[r * Db + c] = t == 0 ? uniform()
                                        : 0.5f * hs[(r - 1) * Db + c] + 0.3f * ha[(r - 1) * Da + c % Da] + 0.4f * uniform();
            for (int c = 0; c < Da; ++c) ha[r * Da + c] = 0.7f * hs[r * Db + c] - 0.5f * hs[r * Db + c + 1];
            if (t + 1 < T) hw[w++] = (int32_t)r;
        }

    float *dp, *dg, *dm, *dv, *dstates, *dactions, *dloss;
    int32_t* dwin;
    void* dws;
    CHECK_HIP(hipMalloc((void**)&dp, nf * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dg, nf * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dm, nf * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dv, nf * sizeof(float)));
    CHECK_HIP(hipMalloc(&dws, wsb));
    CHECK_HIP(hipMalloc((void**)&dstates, n_rows * Db * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dactions, n_rows * Da * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dwin, n_win * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&dloss, 5 * sizeof(float)));
    CHECK_HIP(hipMemcpy(dp, hp, nf * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(dg, 0, nf * sizeof(float)));
    CHECK_HIP(hipMemset(dm, 0, nf * sizeof(float)));
    CHECK_HIP(hipMemset(dv, 0, nf * sizeof(float)));
    CHECK_HIP(hipMemset(dws, 0, wsb));
    CHECK_HIP(hipMemcpy(dstates, hs, n_rows * Db * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dactions, ha, n_rows * Da * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dwin, hw, n_win * sizeof(int32_t), hipMemcpyHostToDevice));

    pvae_ctx* ctx = NULL;
    CHECK(pvae_create(&cfg, &ctx));
    CHECK(pvae_bind_arenas(ctx, dp, dg, dm, dv));
    CHECK(pvae_bind_workspace(ctx, dws, wsb));
    CHECK(pvae_bind_dataset(ctx, dstates, dactions, dwin, n_rows, n_win));

    pvae_step_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.lr = 1e-3; sp.beta1 = 0.9; sp.beta2 = 0.999; sp.adam_eps = 1e-8;
    sp.global_rows = B; sp.rng_seed = 7; sp.loss_kind = PVAE_LOSS_MSE;
    const int steps_per_epoch = (int)(n_win / B);
    float first[2] = {0, 0}, last_[2] = {0, 0}, l[5];
    int t_adam[PVAE_NUM_NETS] = {0, 0, 0};
    for (int phase = 0; phase < 2; ++phase) {             /* PVAE_PHASE_WORLD then PVAE_PHASE_JOINT */
        if (phase == PVAE_PHASE_WORLD) { sp.a_rec_coeff = 0; sp.kl_coeff = 0; sp.s_rec_coeff = 1; sp.cycle_coeff = 0; }
        else { sp.a_rec_coeff = 1; sp.kl_coeff = 1e-3f; sp.s_rec_coeff = 0; sp.cycle_coeff = 1e-3f; }
        for (int epoch = 0; epoch < 6; ++epoch) {
            double mean = 0;
            for (int s = 0; s < steps_per_epoch; ++s) {
                if (phase == PVAE_PHASE_WORLD) sp.adam_t[PVAE_NET_WM] = ++t_adam[PVAE_NET_WM];
                else { sp.adam_t[PVAE_NET_TE] = ++t_adam[PVAE_NET_TE]; sp.adam_t[PVAE_NET_MD] = ++t_adam[PVAE_NET_MD]; }
                sp.rng_offset = (uint64_t)(phase * 100000 + epoch * 1000 + s) * 65536u;
                CHECK(pvae_train_step(ctx, phase, (int64_t)s * B, B, &sp, NULL, dloss, NULL));
                CHECK_HIP(hipMemcpy(l, dloss, sizeof(l), hipMemcpyDeviceToHost));
                for (int k = 0; k < 5; ++k)
                    if (!isfinite(l[k])) { fprintf(stderr, "non-finite loss term %d\n", k); return 4; }
                mean += phase == PVAE_PHASE_WORLD ? l[3] : l[1];        /* loss_s / loss_a */
            }
            mean /= steps_per_epoch;
            if (epoch == 0) first[phase] = (float)mean;
            last_[phase] = (float)mean;
            printf("%s epoch %d: %s = %.6f\n", phase == PVAE_PHASE_WORLD ? "world" : "joint", epoch,
                   phase == PVAE_PHASE_WORLD ? "world-model MSE" : "action MSE", mean);
        }
    }
    /* The consumer of the trained weights (rmt:742-771 at B = 1; envs/rllib_env_imitation.py:215-266): one observation through the
     * per-layer launches (device buffers, stream), then through the call-persistent rollout server (host pointers in and out, no
     * launch per call) -- the two actions must be the same bits. */
    float obs[2 * 23], a_launch[7], a_served[7], *dobs, *dact, *dz;
    for (int c = 0; c < 2 * Db; ++c) obs[c] = hs[c];                    /* rows 0 and 1 of the demonstrations: [s_0 | s_1] */
    CHECK_HIP(hipMalloc((void**)&dobs, sizeof(obs)));
    CHECK_HIP(hipMalloc((void**)&dact, sizeof(a_launch)));
    CHECK_HIP(hipMalloc((void**)&dz, 8 * sizeof(float)));
    CHECK_HIP(hipMemcpy(dobs, obs, sizeof(obs), hipMemcpyHostToDevice));
    CHECK(pvae_infer(ctx, dobs, 1, NULL, 1, 7, 4242, dact, NULL, dz, NULL));
    CHECK_HIP(hipMemcpy(a_launch, dact, sizeof(a_launch), hipMemcpyDeviceToHost));
    CHECK(pvae_rollout_server_start(ctx, 50.0, 30.0, 0));
    CHECK(pvae_rollout_server_infer(ctx, obs, 1, 7, 4242, 0, a_served, NULL, NULL, 1000.0));
    double us[200], med;
    CHECK(pvae_rollout_server_selfbench(ctx, obs, 1, 200, us));
    CHECK(pvae_rollout_server_stop(ctx));
    for (int i = 1; i < 200; ++i) { double v = us[i]; int j = i; while (j > 0 && us[j - 1] > v) { us[j] = us[j - 1]; --j; } us[j] = v; }
    med = us[100];
    if (memcmp(a_launch, a_served, sizeof(a_launch)) != 0) { fprintf(stderr, "served action differs from the launched one\n"); return 6; }
    for (int c = 0; c < Da; ++c) if (!isfinite(a_served[c])) { fprintf(stderr, "non-finite action\n"); return 6; }
    printf("rollout: served action == launched action (bit for bit), %.1f us host observation -> host action\n", med);
    pvae_destroy(ctx);
    if (!(last_[0] < 0.7f * first[0])) { fprintf(stderr, "world-model loss did not fall: %g -> %g\n", first[0], last_[0]); return 5; }
    if (!(last_[1] < 0.9f * first[1])) { fprintf(stderr, "action loss did not fall: %g -> %g\n", first[1], last_[1]); return 5; }
    printf("ok world %.5f -> %.5f, joint %.5f -> %.5f\n", first[0], last_[0], first[1], last_[1]);
    return 0;
}

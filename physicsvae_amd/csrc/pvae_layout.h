// pvae_layout.h -- host-side layout arithmetic shared by the API and the tests:
// where every Linear of the three trainable stacks lives in the flat fp32 arena, and how
// the workspace is carved.  Pure C++ (no HIP), so the layout queries work without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/pvae.h"

namespace pvae {

inline int pad64(int x) { return (x + 63) / 64 * 64; }
inline int pad32(int x) { return (x + 31) / 32 * 32; }

struct Layer {
    int net, index, n_in, n_out, ld, n_out_pad;
    int64_t w_off, b_off;
    bool last;
    int act = 0;                  // act_apply / act_grad code of this layer's output: 0 linear, 1 + PVAE_ACT_* otherwise
    int col0 = 0;                 // first layer on an input subset: the checkpoint tensor is columns [col0, col0 + n_in) of the
                                  // block's rows (ld and the kernels' K stay those of the full-width input)
};

struct NetLayout {
    std::vector<Layer> layers;
    int64_t off = 0, count = 0;   // segment of the arena
    int n_in = 0, n_out = 0;
};

struct Layout {
    pvae_config cfg{};
    NetLayout net[PVAE_NUM_NETS];
    int64_t arena_floats = 0;
    bool ok = false;
    const char* why = "";
};

// order of the stacks in the arena (and of pvae_layer's enumeration)
constexpr int kArenaOrder[PVAE_NUM_NETS] = {PVAE_NET_TE, PVAE_NET_MD, PVAE_NET_MH, PVAE_NET_PR, PVAE_NET_WM};

inline Layout make_layout(const pvae_config& c) {
    Layout L;
    L.cfg = c;
    if (c.dim_body <= 0 || c.dim_action <= 0 || c.latent <= 0) { L.why = "dims must be positive"; return L; }
    if (c.te_width <= 0 || c.md_width <= 0 || c.wm_width <= 0 || c.te_depth <= 0 || c.md_depth <= 0 ||
        c.wm_depth <= 0) { L.why = "width/depth must be positive"; return L; }
    if (c.te_depth > 15 || c.md_depth > 15 || c.wm_depth > 15) { L.why = "depth > 15 unsupported"; return L; }
    if (c.max_batch <= 0 || c.max_batch > 65536) { L.why = "max_batch out of range"; return L; }
    if (c.lookahead < 1 || c.lookahead > 64) { L.why = "lookahead must be in [1, 64]"; return L; }
    if (c.prior_kind < 0 || c.prior_kind > PVAE_PRIOR_NONE) { L.why = "unknown prior_kind"; return L; }
    if (c.act_kind < 0 || c.act_kind > PVAE_ACT_ELU) { L.why = "unknown act_kind"; return L; }
    for (int n = 0; n < PVAE_NUM_NETS; ++n)
        for (int i = 0; i < 16; ++i) {
            if (c.layer_width[n][i] < 0) { L.why = "negative layer_width"; return L; }
            if (c.layer_act[n][i] < 0 || c.layer_act[n][i] > 1 + PVAE_ACT_LINEAR) { L.why = "unknown layer_act"; return L; }
        }
    if (c.prior_kind != PVAE_PRIOR_ZERO_MEAN && c.lookahead != 1) {
        L.why = "latent priors other than normal_zero_mean_one_std need lookahead == 1";
        return L;
    }
    if (c.te_inputs < 0 || c.te_inputs > 3 || c.md_inputs < 0 || c.md_inputs > 3) { L.why = "te_inputs / md_inputs: PVAE_INPUT_* bits"; return L; }
    const bool learned = c.prior_kind == PVAE_PRIOR_STATE_MEAN;
    if (learned && (c.pr_width <= 0 || c.pr_depth <= 0 || c.pr_depth > 15)) { L.why = "prior stack width/depth out of range"; return L; }
    const int Db = c.dim_body, Da = c.dim_action, Z = c.latent;
    const bool helper = c.mh_depth > 0;
    if (c.mh_depth < 0 || c.mh_depth > 15) { L.why = "helper depth out of range"; return L; }
    if (helper && (c.mh_width <= 0 || !(c.mh_range > 0.0f))) { L.why = "helper: mh_width and mh_range must be positive (rmt:672-673)"; return L; }
    // rmt:638-644 (618-621: Z outputs on the hypersphere), 646-668, 682-689, 627-635, 670-680
    const int ins[PVAE_NUM_NETS] = {2 * Db, Db + Z, Db + Da, Db, Db + Z};
    const int outs[PVAE_NUM_NETS] = {c.prior_kind >= PVAE_PRIOR_HYPERSPHERE ? Z : 2 * Z, Da, Db, Z, Da};
    const int widths[PVAE_NUM_NETS] = {c.te_width, c.md_width, c.wm_width, c.pr_width, c.mh_width};
    const int depths[PVAE_NUM_NETS] = {c.te_depth, c.md_depth, c.wm_depth, c.pr_depth, c.mh_depth};
    int64_t off = 0;
    for (int n : kArenaOrder) {
        NetLayout& N = L.net[n];
        N.off = off;
        if (n == PVAE_NET_PR && !learned) continue;          // no such stack: an empty segment
        if (n == PVAE_NET_MH && !helper) continue;
        N.n_in = ins[n];
        N.n_out = outs[n];
        int prev = ins[n];
        // input subsets (rmt:607-613, 646-653): the window of the full-width first layer that the checkpoint tensor is
        int win0 = 0, win = ins[n];
        const int sel = n == PVAE_NET_TE ? c.te_inputs : ((n == PVAE_NET_MD || n == PVAE_NET_MH) ? c.md_inputs : 0);
        if (sel == PVAE_INPUT_BODY) win = Db;
        else if (sel == PVAE_INPUT_TASK) { win0 = Db; win = ins[n] - Db; }
        for (int i = 0; i <= depths[n]; ++i) {
            Layer l;
            l.net = n; l.index = i; l.n_in = i == 0 ? win : prev;
            l.col0 = i == 0 ? win0 : 0;
            l.last = (i == depths[n]);
            l.n_out = l.last ? outs[n] : (c.layer_width[n][i] > 0 ? c.layer_width[n][i] : widths[n]);
            const int act_pub = l.last ? (n == PVAE_NET_MH ? PVAE_ACT_TANH : PVAE_ACT_LINEAR)      // (rmt:672: the helper ends in tanh)
                                       : (c.layer_act[n][i] > 0 ? c.layer_act[n][i] - 1 : c.act_kind);
            l.act = act_pub == PVAE_ACT_LINEAR ? 0 : act_pub + 1;
            l.ld = pad64(prev);
            l.n_out_pad = pad64(l.n_out);
            // (the fused backward launches carry row strides in 16 bits: pvae_gemm.h ga_packable)
            if (l.ld >= 65536 || l.n_out_pad >= 65536) { L.why = "layer wider than 65535 (padded) unsupported"; return L; }
            l.w_off = off; off += (int64_t)l.n_out_pad * l.ld;
            l.b_off = off; off += l.n_out_pad;
            N.layers.push_back(l);
            prev = l.n_out;
        }
        N.count = off - N.off;
    }
    L.arena_floats = off;
    L.ok = true;
    return L;
}

// Workspace carving (all offsets in floats, every buffer 64-float aligned).
//
// lookahead L > 1 (tpv:367-428): every panel holds `slots` time-step blocks stacked along the
// row axis, block s at rows [s*rows_pad, (s+1)*rows_pad) of the CURRENT batch (rows_pad =
// rows rounded up to 32).  TE/MD: one block per step.  WM: blocks [0, L) are the invocations
// with the demonstrated action (tpv:411-414), blocks [L, 2L) those with the decoder's action
// (rmt:758, the state fed to the next step).  Stacking is what lets ONE weight-gradient
// launch per layer contract over all steps (K = slots * rows_pad).
struct NetWork {
    std::vector<int64_t> act;   // act[i]: output of layer i  [slots*Bp][n_out_pad_i]
    std::vector<int64_t> dz;    // dz[i]: grad wrt pre-activation of layer i, same shape
    int64_t in = 0, d_in = 0;   // input panel [slots*Bp][ld0] and its gradient
    int slots = 1;
};

struct Workspace {
    int Bp = 0;                 // rows allocated per block (max_batch padded to 32)
    int L = 1;                  // lookahead
    NetWork net[PVAE_NUM_NETS];
    int64_t s2 = 0, act_t = 0;  // targets: next state [L*Bp][pad64(Db)], action [L*Bp][pad64(Da)]
    int64_t eps = 0;            // eps actually used [L*Bp][Z]
    // second set of staging panels (lookahead 1 only): the gather of minibatch n+1 is written here
    // by tail blocks of step n's last launch, then the two sets swap roles (pvae_train_step_prefetch)
    int64_t alt_in[PVAE_NUM_NETS] = {};
    int64_t alt_s2 = 0, alt_act_t = 0;
    int64_t loss_part = 0;      // [5][kLossParts] partial sums
    int64_t obs_keep = 0;       // [4][2*Db] the observation rows of the last <= 4-row rollout call (pvae_infer)
    int64_t z_side = 0;         // motor_decoder_inputs = ["body"] only: where the sampler's z goes instead of the decoder's input panel
    int64_t zero = 0;           // 64 floats that nothing ever writes: what a gathered first-layer operand reads for "no source"
    int64_t total_floats = 0;
};

constexpr int kLossParts = 8192;   // max workgroups contributing to one loss term

inline Workspace make_workspace(const Layout& L) {
    Workspace W;
    W.Bp = pad32(L.cfg.max_batch);
    W.L = L.cfg.lookahead;
    const int64_t T = W.L;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t o = off; off += (n + 63) / 64 * 64; return o; };
    for (int n = 0; n < PVAE_NUM_NETS; ++n) {
        const NetLayout& N = L.net[n];
        if (N.layers.empty()) continue;
        const int64_t slots = (T > 1 && n == PVAE_NET_WM) ? 2 * T : T;
        W.net[n].slots = (int)slots;
        // (the helper reads the decoder's input panel: PVAE_NET_MD < PVAE_NET_MH, so that one is carved already)
        W.net[n].in = n == PVAE_NET_MH ? W.net[PVAE_NET_MD].in : take(slots * W.Bp * N.layers[0].ld);
        W.net[n].d_in = take(slots * W.Bp * N.layers[0].ld);
        for (const Layer& l : N.layers) {
            W.net[n].act.push_back(take(slots * W.Bp * l.n_out_pad));
            W.net[n].dz.push_back(take(slots * W.Bp * l.n_out_pad));
        }
    }
    W.s2 = take(T * W.Bp * pad64(L.cfg.dim_body));
    W.act_t = take(T * W.Bp * pad64(L.cfg.dim_action));
    W.eps = take(T * W.Bp * L.cfg.latent);
    for (int n = 0; n < PVAE_NUM_NETS; ++n)
        if (!L.net[n].layers.empty())
            W.alt_in[n] = n == PVAE_NET_MH ? W.alt_in[PVAE_NET_MD] : take((int64_t)W.Bp * L.net[n].layers[0].ld);
    W.alt_s2 = take((int64_t)W.Bp * pad64(L.cfg.dim_body));
    W.alt_act_t = take((int64_t)W.Bp * pad64(L.cfg.dim_action));
    W.loss_part = take(5 * kLossParts);
    W.obs_keep = take(4 * 2 * (int64_t)L.cfg.dim_body);
    if (L.cfg.md_inputs == PVAE_INPUT_BODY)
        W.z_side = take((int64_t)W.net[PVAE_NET_MD].slots * W.Bp * L.net[PVAE_NET_MD].layers[0].ld);
    W.zero = take(64);
    W.total_floats = off;
    return W;
}

}  // namespace pvae

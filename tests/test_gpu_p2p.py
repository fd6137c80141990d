"""The peer-mapped gradient exchange (PVAE_EXCHANGE_P2P; SURVEY.md section 8e's direct all-pairs reduce-scatter +
all-gather) on ONE GPU: 2 and 4 processes share cuda:0, map one another's gradient / parameter arenas and flag
blocks with hipIpcOpenMemHandle, and exchange inside kernels -- every "peer" is another process on the same
device, which exercises the mapping, the cross-process flags, the ordering and the arithmetic (not the links).
The reference has no counterpart (tm:131-161 is one process)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib, torch
root = sys.argv[1]; out = sys.argv[2]; mode = sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from physicsvae_amd import parallel, _lib
rank, world, _ = parallel.init_from_env(backend="gloo")
import torch.distributed as dist
from oracle import refpath as R
from util import make_trainer
from physicsvae_amd.engine import make_step_params
arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")          # 117 windows
per_gpu = int(sys.argv[3])
look = int(os.environ.get("P2P_TEST_LOOKAHEAD", "1"))
extra = {"lookahead": look} if look > 1 else {}
if os.environ.get("P2P_TEST_SHUFFLE") == "1":
    extra["shuffle_data"] = True
    torch.manual_seed(100 + rank)              # every rank would draw ANOTHER order: rank 0's is what all must use
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(arch, data, per_gpu, m_world=1, device="cuda:0", eps_fn=R.eps_stream(2, 8),
                      extra=extra or None)
tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, 1), 3))
eng = tr.engine
if "P2P_TEST_DELAY_US" in os.environ:          # a spin kernel in front of every exchange launch, a different one per rank
    delay = int(os.environ["P2P_TEST_DELAY_US"].split(",")[rank])
    eng.comm_config(float(os.environ.get("PVAE_DP_BUCKET_MB", "0")), delay)
res = {}
if mode == "train":
    if os.environ.get("PVAE_DP_EXCHANGE") in ("p2p", "p2p_push"):
        assert eng.has_p2p and not eng.has_comm and eng.p2p_status()[:2] == (rank, world)
    losses = [tr.train()["mean_train_loss"] for _ in range(2)]
    res = {"sd": {k: v.cpu() for k, v in tr.model.state_dict().items()}, "losses": losses,
           "steps": dict(tr.optimizer.net_steps), "timeouts": eng.p2p_status()[2] if eng.has_p2p else 0,
           "report": {"chosen": getattr(tr, "dp_exchange_chosen", None), "candidates": tr.dp_exchange_report},
           "phases_calibrated": list(tr.dp_exchange_reports), "replica_checks": sorted(tr._replicas_checked)}
elif mode == "kernel":
    # the exchange launch itself: known gradients per rank, slice owners sum in RANK ORDER, Adam, push
    assert eng.has_p2p
    te_off, te_cnt = eng.segments[_lib.NET_TE]
    ok = True
    for step in (1, 2, 3, 4, 5):
        if step == 4:
            if os.environ.get("P2P_TEST_REOPEN") != "1":
                break
            # the set-up torn down and made again (what a calibration does when it moves from one form to the next
            # after a failed attach): flag epochs of the closed set-up must not satisfy the new one's waits
            torch.cuda.synchronize(); dist.barrier()
            eng.p2p_close()
            assert not eng.has_p2p and eng.p2p_status()[1] == 0
            dist.barrier()
            assert tr.dp.attach_p2p(eng, os.environ["PVAE_DP_EXCHANGE"]) and eng.has_p2p
        g = torch.Generator(device="cuda").manual_seed(100 * step + rank)
        eng.grads.copy_(torch.randn(eng.grads.numel(), generator=g, device="cuda") * 1e-2)
        torch.cuda.synchronize(); dist.barrier()
        p0, m0, v0 = eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()
        mine = eng.grads.clone()
        sp = make_step_params(lr=5e-4, adam_t=(step, step, step), global_rows=32)
        eng.p2p_exchange(_lib.NET_TE, te_off, te_cnt, sp)
        torch.cuda.synchronize()
        got_p, got_m, got_v = eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()
        # expected: every rank's gradient (gathered over gloo), summed in rank order, flat Adam on the whole slice
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        total = parts[0].clone()
        for q in range(1, world):
            total += parts[q]
        eng.params.copy_(p0); eng.exp_avg.copy_(m0); eng.exp_avg_sq.copy_(v0)
        eng.grads.copy_(total)
        eng.adam_segment(_lib.NET_TE, te_off, te_cnt, sp)
        torch.cuda.synchronize()
        sl = slice(te_off, te_off + te_cnt)
        n4 = te_cnt // 4
        S = (n4 + world - 1) // world
        own = slice(te_off + 4 * rank * S, te_off + 4 * min((rank + 1) * S, n4))
        ok = ok and torch.equal(got_p[sl], eng.params[sl])                      # all slices, whoever owned them
        ok = ok and torch.equal(got_m[own], eng.exp_avg[own]) and torch.equal(got_v[own], eng.exp_avg_sq[own])
        ok = ok and torch.equal(got_p[: te_off], p0[: te_off]) and torch.equal(got_p[te_off + te_cnt:], p0[te_off + te_cnt:])
        # carry the exchanged state into the next round.  Each rank holds valid moments for ITS slice only (as the
        # mode defines), so the expectation above needs them assembled from their owners first.
        eng.params.copy_(got_p)
        for dst, src in ((eng.exp_avg, got_m), (eng.exp_avg_sq, got_v)):
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            for q in range(world):
                oq = slice(te_off + 4 * q * S, te_off + 4 * min((q + 1) * S, n4))
                dst[oq] = parts[q][oq]
        torch.cuda.synchronize(); dist.barrier()
    res = {"ok": bool(ok), "timeouts": eng.p2p_status()[2]}
elif mode == "resume":
    # interrupted after `stop` epochs under the peer-mapped exchange, resumed from trainer_state.pt written after the
    # collective gather of the moments; `full` = the uninterrupted run
    stop = int(os.environ["P2P_TEST_STOP"])
    ck_dir = sys.argv[2] + "_ck"
    if os.environ.get("P2P_TEST_ROLE") == "first":
        for _ in range(stop):
            tr.train()
        assert tr.save_trainer_state(os.path.join(ck_dir, "never.pt")) is None      # not gathered: refused
        assert tr.gather_moments() is True
        if rank == 0:
            os.makedirs(ck_dir, exist_ok=True)
            torch.save(tr.model.portable_state_dict(), os.path.join(ck_dir, "model.pth"))
            assert tr.save_trainer_state(os.path.join(ck_dir, "trainer_state.pt")) is not None
        dist.barrier()
        losses = [tr.train()["mean_train_loss"] for _ in range(2)]
    else:
        tr.model.load_state_dict(torch.load(os.path.join(ck_dir, "model.pth"), map_location="cpu"))
        tr.load_trainer_state(os.path.join(ck_dir, "trainer_state.pt"))
        losses = [tr.train()["mean_train_loss"] for _ in range(2)]
    own = []
    for net, phase in ((_lib.NET_WM, _lib.PHASE_WORLD), (_lib.NET_TE, _lib.PHASE_JOINT), (_lib.NET_MD, _lib.PHASE_JOINT)):
        own += [(o, c) for o, c, rep in eng.owned_slices(phase, net)]
    res = {"sd": {k: v.cpu() for k, v in tr.model.state_dict().items()}, "losses": losses,
           "m_own": torch.cat([eng.exp_avg[o: o + c] for o, c in own]).cpu(), "steps": dict(tr.optimizer.net_steps),
           "timeouts": eng.p2p_status()[2]}
elif mode == "timeout":
    # rank 0 enters an exchange its peer never joins: the wait gives up (PVAE_P2P_TIMEOUT_MS), no hang
    assert eng.has_p2p
    dist.barrier()
    if rank == 0:
        sp = make_step_params(lr=5e-4, global_rows=32)
        eng.p2p_exchange(_lib.NET_TE, eng.segments[_lib.NET_TE][0], eng.segments[_lib.NET_TE][1], sp)
    res = {"timeouts": eng.p2p_status()[2]}
torch.save(res, out + ".%d" % rank)
if world > 1:
    dist.barrier()
print("DONE", rank)
'''


def _run(tmp_path, world, per_gpu, tag, mode="train", port="29561", **extra_env):
    script = tmp_path / "p2p_worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / tag)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, out, str(per_gpu), mode],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(out + ".%d" % r) for r in range(world)]


@pytest.mark.parametrize("form", ["p2p", "p2p_push"])
def test_p2p_two_ranks_equal_the_allreduce_exchange_bit_for_bit(tmp_path, form):
    """Two processes on one GPU, two epochs across the phase switch, ragged tail with an empty shard: replicas end
    bit-identical, and equal what the all-reduce + replicated-Adam exchange produces on the same schedule (with two
    ranks every sum is g0 + g1 whoever forms it).  Both forms: owners PULL their slice from the peers' gradient arenas,
    or every rank PUSHES its contributions into the owners' staging buffers (remote writes only)."""
    p2p = _run(tmp_path, 2, 16, "p2p", PVAE_DP_EXCHANGE=form)
    plain = _run(tmp_path, 2, 16, "plain", port="29562", PVAE_DP_EXCHANGE="default")
    a, b = p2p
    assert a["timeouts"] == 0 and b["timeouts"] == 0
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k
        assert torch.equal(a["sd"][k], plain[0]["sd"][k]), k
    assert a["losses"] == plain[0]["losses"] and a["steps"] == plain[0]["steps"]


def test_torch_transport_with_an_empty_shard_issues_its_peers_collectives(tmp_path):
    """117 windows, 24 per rank, two ranks: the last global minibatch leaves rank 1 with NO rows, in the world phase and
    in the joint phase (two stacks = two buckets; with PVAE_DP_BUCKET_MB one per layer).  On the torch.distributed
    transport the empty rank launches nothing, so it takes the sequence of collectives from the backward PLAN
    (pvae_backward_plan, torch_models.dp_buckets) -- same sizes, same order as its peer issues behind its stages, or gloo
    would pair a bucket with a whole segment.  Replicas bit-identical, equal to the peer-mapped exchange (two ranks: every
    sum is g0 + g1), whatever the bucket size."""
    p2p = _run(tmp_path, 2, 24, "p2p24", port="29591", PVAE_DP_EXCHANGE="p2p")
    plain = _run(tmp_path, 2, 24, "plain24", port="29592", PVAE_DP_EXCHANGE="default")
    small = _run(tmp_path, 2, 24, "small24", port="29593", PVAE_DP_EXCHANGE="default", PVAE_DP_BUCKET_MB="0.01")
    for k in p2p[0]["sd"]:
        assert torch.equal(plain[0]["sd"][k], plain[1]["sd"][k]), k
        assert torch.equal(plain[0]["sd"][k], p2p[0]["sd"][k]), k
        assert torch.equal(plain[0]["sd"][k], small[0]["sd"][k]) and torch.equal(small[0]["sd"][k], small[1]["sd"][k]), k
    assert plain[0]["losses"] == p2p[0]["losses"] == small[0]["losses"] and plain[0]["steps"] == p2p[0]["steps"]


def test_shuffled_epochs_data_parallel_use_rank_zeros_order(tmp_path):
    """shuffle_data with two ranks whose default generators are seeded differently: every pass uses rank 0's permutation
    (broadcast), each rank taking its slice of it -- replicas bit-identical, every window seen once per epoch (the losses
    equal those of the same run on the torch.distributed transport), and not the sequential run's."""
    a = _run(tmp_path, 2, 16, "shuf", port="29594", PVAE_DP_EXCHANGE="p2p", P2P_TEST_SHUFFLE="1")
    b = _run(tmp_path, 2, 16, "shufplain", port="29595", PVAE_DP_EXCHANGE="default", P2P_TEST_SHUFFLE="1")
    seq = _run(tmp_path, 2, 16, "seq", port="29596", PVAE_DP_EXCHANGE="p2p")
    for k in a[0]["sd"]:
        assert torch.equal(a[0]["sd"][k], a[1]["sd"][k]) and torch.equal(a[0]["sd"][k], b[0]["sd"][k]), k
    assert a[0]["losses"] == b[0]["losses"] and a[0]["losses"] != seq[0]["losses"]


def test_p2p_bucketed_overlapped_equals_in_line(tmp_path):
    """The same exchange cut into per-layer buckets on the library's exchange stream (event hand-offs, several
    exchange launches in flight per step) gives the in-line result bit for bit."""
    inline = _run(tmp_path, 2, 16, "inline", PVAE_DP_EXCHANGE="p2p")
    bucketed = _run(tmp_path, 2, 16, "bucketed", port="29563", PVAE_DP_EXCHANGE="p2p", PVAE_DP_BUCKET_MB="0.01")
    assert all(r["timeouts"] == 0 for r in inline + bucketed)
    for k in inline[0]["sd"]:
        assert torch.equal(inline[0]["sd"][k], bucketed[0]["sd"][k]), k
        assert torch.equal(bucketed[0]["sd"][k], bucketed[1]["sd"][k]), k


def test_p2p_ranks_that_arrive_at_different_times(tmp_path):
    """Skew between the ranks (a 0 / 700 us spin kernel in front of every exchange launch of rank 0 / rank 1, in line and
    with per-layer buckets on the exchange stream): the early rank waits inside its exchange launch for the late one's
    "gradient final" flag and for its "done" flag; parameters end bit-identical to the run without skew."""
    base = _run(tmp_path, 2, 16, "noskew", PVAE_DP_EXCHANGE="p2p")
    for tag, port, form, extra in (("skew", "29565", "p2p", {}), ("skewb", "29566", "p2p", {"PVAE_DP_BUCKET_MB": "0.01"}),
                                   ("skewp", "29568", "p2p_push", {}), ("skewpb", "29569", "p2p_push", {"PVAE_DP_BUCKET_MB": "0.01"})):
        got = _run(tmp_path, 2, 16, tag, port=port, PVAE_DP_EXCHANGE=form, P2P_TEST_DELAY_US="700,0" if "p" in tag[4:] else "0,700", **extra)
        assert all(r["timeouts"] == 0 for r in got)
        for k in base[0]["sd"]:
            assert torch.equal(base[0]["sd"][k], got[0]["sd"][k]) and torch.equal(got[0]["sd"][k], got[1]["sd"][k]), (tag, k)


def test_exchange_chosen_by_measurement_leaves_no_trace(tmp_path):
    """`dp_exchange = "auto"`: the first training epoch times every exchange form available (here: the two peer-mapped
    forms; RCCL needs one GPU per rank) from a snapshot of parameters and moments and restores it, so the run that
    follows equals a run with that form chosen by hand bit for bit -- calibration steps leave no trace."""
    auto = _run(tmp_path, 2, 16, "auto", PVAE_DP_EXCHANGE="auto")
    hand = _run(tmp_path, 2, 16, "hand", port="29583", PVAE_DP_EXCHANGE="p2p")
    assert all(r["timeouts"] == 0 for r in auto)
    for k in auto[0]["sd"]:
        assert torch.equal(auto[0]["sd"][k], auto[1]["sd"][k]), k
        assert torch.equal(auto[0]["sd"][k], hand[0]["sd"][k]), k
    assert auto[0]["losses"] == hand[0]["losses"] and auto[0]["steps"] == hand[0]["steps"]
    rep = auto[0]["report"]
    assert rep["chosen"] in ("p2p", "p2p_push")
    assert all("skipped" in rep["candidates"][f] for f in ("inline", "bucketed", "sharded"))
    assert all(rep["candidates"][f]["replicas_identical"] and rep["candidates"][f]["us_per_step"] > 0 for f in ("p2p", "p2p_push"))


def test_unset_dp_exchange_means_auto_and_calibrates_each_phase(tmp_path):
    """With more than one rank an unset `dp_exchange` IS "auto" (round 4): the first epoch of the world phase and the
    first epoch of the joint phase each calibrate (the phases have different amounts of work to hide an exchange
    behind), the run equals the hand-chosen form bit for bit, and the replicas are checked after each phase's first
    epoch whatever chose the form."""
    if "PVAE_DP_EXCHANGE" in os.environ:
        pytest.skip("PVAE_DP_EXCHANGE is set in the environment")
    auto = _run(tmp_path, 2, 16, "unset", port="29586")
    hand = _run(tmp_path, 2, 16, "hand2", port="29587", PVAE_DP_EXCHANGE="p2p")
    assert all(r["timeouts"] == 0 for r in auto)
    assert auto[0]["report"]["chosen"] in ("p2p", "p2p_push") and sorted(auto[0]["phases_calibrated"]) == [0, 1]
    assert auto[0]["replica_checks"] == [0, 1] and hand[0]["replica_checks"] == [0, 1]
    for k in auto[0]["sd"]:
        assert torch.equal(auto[0]["sd"][k], auto[1]["sd"][k]), k
        assert torch.equal(auto[0]["sd"][k], hand[0]["sd"][k]), k
    assert auto[0]["losses"] == hand[0]["losses"]


@pytest.mark.parametrize("stop", [1, 2])
def test_trainer_state_resume_under_the_peer_mapped_exchange(tmp_path, stop):
    """Under the sharded / peer-mapped exchanges a rank keeps Adam moments for its own slices only, so
    `trainer_state.pt` needs the collective `gather_moments` first (SUM all-reduce of the moments masked to what each
    rank owns: exact).  A two-rank run interrupted before / after the phase switch and resumed from that file
    continues bit for bit: epoch losses, weights, Adam counters, and the moments each rank owns."""
    first = _run(tmp_path, 2, 16, "res", mode="resume", port="29584", PVAE_DP_EXCHANGE="p2p", P2P_TEST_STOP=str(stop),
                 P2P_TEST_ROLE="first")
    again = _run(tmp_path, 2, 16, "res", mode="resume", port="29585", PVAE_DP_EXCHANGE="p2p", P2P_TEST_STOP=str(stop),
                 P2P_TEST_ROLE="second")
    for a, b in zip(first, again):
        assert a["timeouts"] == 0 and b["timeouts"] == 0
        assert a["losses"] == b["losses"] and a["steps"] == b["steps"]
        assert torch.equal(a["m_own"], b["m_own"])
        for k in a["sd"]:
            assert torch.equal(a["sd"][k], b["sd"][k]), k


def test_p2p_with_a_lookahead_unroll(tmp_path):
    """lookahead 2 (stacked time steps, weight gradients paired with step 0's input-gradient launches) under the
    peer-mapped exchange: replicas bit-identical and equal to the all-reduce exchange of the same schedule."""
    p2p = _run(tmp_path, 2, 16, "p2pL2", PVAE_DP_EXCHANGE="p2p", P2P_TEST_LOOKAHEAD="2")
    plain = _run(tmp_path, 2, 16, "plainL2", port="29567", P2P_TEST_LOOKAHEAD="2", PVAE_DP_EXCHANGE="default")
    assert all(r["timeouts"] == 0 for r in p2p)
    for k in p2p[0]["sd"]:
        assert torch.equal(p2p[0]["sd"][k], p2p[1]["sd"][k]), k
        assert torch.equal(p2p[0]["sd"][k], plain[0]["sd"][k]), k
    assert p2p[0]["losses"] == plain[0]["losses"]


@pytest.mark.parametrize("form", ["p2p", "p2p_push"])
def test_p2p_four_ranks_replicas_identical_and_match_one_process(tmp_path, form):
    """Four processes on one GPU (global batch 32 = 4 x 8): replicas bit-identical; against one process with the
    global batch the parameters agree to fp32 summation order."""
    dp = _run(tmp_path, 4, 8, "p2p4", PVAE_DP_EXCHANGE=form)
    single = _run(tmp_path, 1, 32, "single", port="29564")[0]
    assert all(r["timeouts"] == 0 for r in dp)
    for r in dp[1:]:
        for k in dp[0]["sd"]:
            assert torch.equal(dp[0]["sd"][k], r["sd"][k]), k
    assert dp[0]["steps"] == single["steps"]
    assert dp[0]["losses"] == pytest.approx(single["losses"], rel=1e-5)
    for k, v in single["sd"].items():
        if k.startswith("_value_branch"):
            continue
        err = float((dp[0]["sd"][k] - v).norm() / (v.norm() + 1e-30))
        assert err < 2e-3, (k, err)


@pytest.mark.parametrize("form", ["p2p", "p2p_push"])
def test_p2p_eight_ranks_replicas_identical_and_match_one_process(tmp_path, form):
    """BASELINE configs[3]'s rank count on the hardware there is: eight processes on one GPU (global batch 64 = 8 x 8),
    every flag word, slice boundary and template instance of the 8-rank exchange in use.  Replicas bit-identical;
    against one process with the global batch the parameters agree to fp32 summation order."""
    dp = _run(tmp_path, 8, 8, "p2p8", port="29584", PVAE_DP_EXCHANGE=form)
    single = _run(tmp_path, 1, 64, "single8", port="29585")[0]
    assert all(r["timeouts"] == 0 for r in dp)
    for r in dp[1:]:
        for k in dp[0]["sd"]:
            assert torch.equal(dp[0]["sd"][k], r["sd"][k]), k
    assert dp[0]["steps"] == single["steps"]
    assert dp[0]["losses"] == pytest.approx(single["losses"], rel=1e-5)
    for k, v in single["sd"].items():
        if k.startswith("_value_branch"):
            continue
        err = float((dp[0]["sd"][k] - v).norm() / (v.norm() + 1e-30))
        assert err < 2e-3, (k, err)


@pytest.mark.parametrize("form", ["p2p", "p2p_push"])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_p2p_exchange_launch_sums_in_rank_order_and_updates_every_replica(tmp_path, world, form):
    """The exchange launch on known gradients: the parameters every rank ends with equal flat Adam on the rank-order
    sum ((g0 + g1) + g2) + g3 bit for bit, over all slices (own slice computed here, the others pushed by their
    owners), moments are updated on the own slice, nothing outside the exchanged segment moves; three rounds over
    the same buffers (flag epochs, ticket reset, stale-cache hazards)."""
    res = _run(tmp_path, world, 8, "k%d" % world, mode="kernel", port=str(29570 + world), PVAE_DP_EXCHANGE=form)
    assert all(r["ok"] and r["timeouts"] == 0 for r in res), res


@pytest.mark.parametrize("form", ["p2p", "p2p_push"])
def test_p2p_closed_and_attached_again_starts_from_clean_flags(tmp_path, form):
    """pvae_p2p_close, then export / open / self-test again in the same processes (three ranks): the exchange launches
    of the second set-up are as exact as the first one's, no wait gives up."""
    res = _run(tmp_path, 3, 8, "reopen", mode="kernel", port="29591", PVAE_DP_EXCHANGE=form, P2P_TEST_REOPEN="1")
    assert all(r["ok"] and r["timeouts"] == 0 for r in res), res


def test_p2p_wait_for_a_missing_peer_gives_up_instead_of_hanging(tmp_path):
    res = _run(tmp_path, 2, 8, "to", mode="timeout", port="29580", PVAE_DP_EXCHANGE="p2p", PVAE_P2P_TIMEOUT_MS="300")
    assert res[0]["timeouts"] >= 1 and res[1]["timeouts"] == 0


def test_p2p_calls_fail_loudly_without_setup():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import refpath as R
    from physicsvae_amd import _lib
    from physicsvae_amd.engine import make_step_params
    from util import make_trainer
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
    data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")
    tr = make_trainer(arch, data, 32, device="cuda")
    eng = tr.engine
    sp = make_step_params(lr=5e-4, global_rows=32)
    with pytest.raises(RuntimeError, match="not open"):
        eng.comm_mode("p2p")
    with pytest.raises(RuntimeError, match="not open"):
        eng.p2p_exchange(_lib.NET_TE, *eng.segments[_lib.NET_TE], sp)
    assert eng.p2p_status() == (0, 0, 0)
    blob = eng.p2p_export()                         # a single rank can open itself: the exchange is then local Adam
    eng.p2p_open(0, 1, [blob])
    eng.p2p_selftest()                              # (no peers: trivially passes)
    eng.comm_mode("p2p_push")
    eng.comm_mode("p2p")
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    sd = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    eps = R.eps_stream(2, 8)(0, (32, 8))
    for phase, world in ((_lib.PHASE_WORLD, True), (_lib.PHASE_JOINT, False)):
        c = R.phase_coeffs(world)
        res = []
        for dp in (False, True):
            tr.model.load_state_dict(sd)
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            out = torch.zeros(5, device="cuda")
            for t in (1, 2, 3):
                spt = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                       s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=32)
                (eng.dp_train_step if dp else eng.train_step)(phase, 32 * (t - 1), 32, spt, eps=eps, loss_out=out)
            res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)                # one rank through the exchange launch == the fused single-GPU step
    assert eng.p2p_status() == (0, 1, 0)
    eng.p2p_close()
    assert eng.p2p_status()[:2] == (0, 0)

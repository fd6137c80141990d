"""Data-parallel sharding of the minibatch schedule: one process per GPU, gradients summed
with an RCCL all-reduce over xGMI (`torch.distributed` backend "nccl" on ROCm; "gloo" in the
CPU tests).

The path shards over independent samples with ONE exchange per optimizer step (SURVEY.md
8e): global minibatch g covers windows [g*Bg, (g+1)*Bg) in the reference's sequential
order, Bg = world * batch_size; rank r takes the contiguous slice starting at r*batch_size.
Every rank scales its loss terms and gradients by 1/global_rows (not by its local row
count), so a plain SUM all-reduce reproduces the reference's batch-mean gradient, including
on the ragged last global batch where shards may be short or empty.  Parameters and Adam
moments are replicated and every rank applies the identical update to the identical
reduced gradient, so replicas stay bit-identical.
"""
import os
import sys

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, rank=0, world=1, group=None, always_reduce=False):
        self.rank, self.world, self.group = int(rank), int(world), group
        # run the staged backward + collective + per-slice Adam path even for a single rank
        # (PVAE_DP_ALWAYS_REDUCE=1): exercises RCCL and its stream ordering on a 1-GPU box
        self.always_reduce = bool(always_reduce) and dist.is_available() and dist.is_initialized()

    @property
    def collective(self):
        return self.world > 1 or self.always_reduce

    @classmethod
    def from_env(cls):
        force = os.environ.get("PVAE_DP_ALWAYS_REDUCE", "0") == "1"
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size(), always_reduce=force)
        return cls(0, 1)

    def global_steps(self, n_windows, batch_size):
        bg = batch_size * self.world
        return (n_windows + bg - 1) // bg

    def global_first(self, g, batch_size):
        return g * batch_size * self.world

    def shard(self, g, n_windows, batch_size):
        """-> (first_window, rows, global_rows) of this rank in global minibatch g."""
        gfirst = self.global_first(g, batch_size)
        gend = min(gfirst + batch_size * self.world, n_windows)
        first = gfirst + self.rank * batch_size
        rows = max(0, min(batch_size, gend - first))
        return (first if rows else 0), rows, gend - gfirst

    def attach(self, engine):
        """Give `engine`'s ctx its own RCCL communicator so the whole data-parallel step runs inside
        the library (`pvae_dp_train_step`: all-reduce stream-ordered with the kernels, no Python
        between launches).  Needs the nccl backend (one GPU per rank); PVAE_DP_TRANSPORT=torch keeps
        torch.distributed as the transport, which is also what gloo runs and any failure to set the
        communicator up fall back to -- a different transport for the same exchange."""
        if not self.collective or engine.has_comm or engine.ctx is None:
            return engine.has_comm
        if os.environ.get("PVAE_DP_TRANSPORT", "rccl") != "rccl" or dist.get_backend(self.group) != "nccl":
            return False
        # Every step below is collective: a rank that hits an error must still take part, then all
        # ranks agree (MIN over a success flag) on ONE transport -- a rank raising on its own would
        # leave the others waiting in the broadcast, or later in mismatched collectives.
        uid, err = None, None
        if self.rank == 0:
            try:
                uid = engine.comm_unique_id()
            except Exception as exc:                               # noqa: BLE001
                err = str(exc)
        box = [uid]
        dist.broadcast_object_list(box, src=0, group=self.group)
        ok = box[0] is not None
        if ok:
            try:
                engine.comm_init(self.rank, self.world, box[0])
            except Exception as exc:                               # noqa: BLE001
                ok, err = False, str(exc)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) != 1:
            if engine.has_comm:
                engine.comm_destroy()
            if self.rank == 0 or err:
                print("[physicsvae_amd] in-library RCCL exchange unavailable (%s); using torch.distributed"
                      % (err or "another rank failed"), file=sys.stderr)
            return False
        return True

    def attach_p2p(self, engine, form="p2p"):
        """Peer-mapped exchange (`dp_exchange = "p2p"`, include/pvae.h PVAE_EXCHANGE_P2P): every rank exports IPC
        handles of its gradient / parameter arenas and flag block, the blobs are all-gathered over whatever
        backend torch.distributed runs (nccl, or gloo when the ranks share one GPU), every rank maps its peers'
        buffers, and `pvae_dp_train_step` then exchanges each bucket with ONE launch, without RCCL.  Collective:
        every rank takes part in every step and all agree (MIN over a success flag) on the outcome."""
        if not self.collective or engine.ctx is None:
            return False
        assert form in ("p2p", "p2p_push")
        if engine.has_p2p:
            engine.comm_mode(form)
            return True
        blob, err = None, None
        try:
            blob = engine.p2p_export()
        except Exception as exc:                                   # noqa: BLE001
            err = str(exc)
        blobs = [None] * self.world
        dist.all_gather_object(blobs, blob, group=self.group)
        ok = all(b is not None for b in blobs)
        if ok:
            try:
                engine.p2p_open(self.rank, self.world, blobs)
            except Exception as exc:                               # noqa: BLE001
                ok, err = False, str(exc)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        head = None
        if int(flag.item()) == 1:
            # every rank has its peers mapped: prove remote write / remote read / flag delivery before the first step
            # (bounded: a mapping that does not reach its peer fails within a second instead of hanging a training step).
            # The arena part of the test writes patterns into the first 1 KB of the LIVE parameter and gradient arenas
            # and restores them itself -- but after a wait that timed out, a late peer's store can land behind that
            # restore.  So the words are kept here too, and put back once every rank has unmapped its peers (below).
            head = (engine.params[:256].clone(), engine.grads[:256].clone())
            try:
                engine.p2p_selftest()
            except Exception as exc:                               # noqa: BLE001
                ok, err = False, str(exc)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) != 1:
            engine.p2p_close()
            if head is not None:
                torch.cuda.synchronize(engine.device)
                dist.barrier(group=self.group)                     # nobody can reach this rank's arenas any more
                engine.params[:256].copy_(head[0]); engine.grads[:256].copy_(head[1])
                # ... and not trusted on its own word: the tested words of the parameter arena travel once more from rank 0
                # (the replicas were identical when the test began), then every rank's WHOLE arena is compared -- a form
                # that failed its self-test must not leave replicas that differ anywhere behind.
                if self.world > 1:
                    t = engine.params[:256].clone()
                    dist.broadcast(t, src=0, group=self.group)
                    engine.params[:256].copy_(t)
                engine.params_changed()
                if self.world > 1 and not _replicas_identical(self, engine):
                    raise RuntimeError("peer-mapped exchange: the self-test failed AND the replicas' parameters differ after "
                                       "its test words were restored -- refusing to train on")
            if self.rank == 0 or err:
                print("[physicsvae_amd] peer-mapped exchange unavailable (%s)" % (err or "another rank failed"), file=sys.stderr)
            return False
        engine.comm_mode(form)
        return True

    def all_reduce(self, tensor):
        if self.collective:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def broadcast(self, tensor, device=None):
        """Rank 0's `tensor` on every rank (a shuffled epoch's sample order: every rank slices the SAME permutation).  A
        host tensor travels as it is over gloo and through `device` over RCCL, which moves device memory only."""
        if self.world <= 1 or not (dist.is_available() and dist.is_initialized()):
            return tensor
        if dist.get_backend(self.group) == "nccl" and not tensor.is_cuda:
            t = tensor.to(device)
            dist.broadcast(t, src=0, group=self.group)
            return t.cpu()
        dist.broadcast(tensor, src=0, group=self.group)
        return tensor


EXCHANGE_FORMS = ("inline", "bucketed", "sharded", "p2p", "p2p_push")


def _set_exchange_form(self, engine, form):
    """Switch the in-library exchange of `engine` to `form` (collective for the peer-mapped forms: every rank calls it with
    the same value).  -> None, or the reason the form is not available here."""
    if form in ("p2p", "p2p_push"):
        if not self.attach_p2p(engine, form):
            return "peer-mapped exchange could not be set up"
        engine.comm_config(0.0)
        return None
    if not engine.has_comm:
        return "no RCCL communicator"
    engine.comm_mode("sharded" if form == "sharded" else "allreduce")
    engine.comm_config(6.0 if form == "bucketed" else 0.0)
    return None


def _replicas_identical(self, engine):
    """Every rank holds bit-identical parameters (checksum over the bit patterns, MAX == MIN over the ranks)."""
    h = engine.params.view(torch.int32).to(torch.int64).sum().reshape(1)
    hi, lo = h.clone(), h.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
    return int(hi.item()) == int(lo.item())


def _all_agree(self, engine, ok):
    """True iff `ok` holds on EVERY rank (MIN over the ranks): the outcome all of them act on."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
    return int(flag.item()) == 1


def _p2p_timeouts(self, engine):
    """Waits of the peer-mapped exchange that gave up, MAX over the ranks (a time-out on one rank concerns all)."""
    bad = torch.tensor([engine.p2p_status()[2] if engine.has_p2p else 0], dtype=torch.int64, device=engine.device)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
    return int(bad.item())


def _autotune_exchange(self, engine, run_steps, steps=24, warm=6, forms=EXCHANGE_FORMS):
    """Choose the exchange form by measurement: under every form available here, `run_steps(n)` (n data-parallel
    optimizer steps, supplied by the caller) is timed from the SAME starting state -- parameters and Adam moments are
    snapshotted first and restored after every candidate and at the end, so the calibration leaves no trace --, the
    time is the maximum over the ranks (so every rank takes the same decision), and a candidate only counts if the
    replicas are bit-identical after its steps and no peer wait gave up.  -> (chosen form | None, report).

    Every decision is collective: whether a candidate ran through is agreed with a MIN over the ranks OUTSIDE the
    try blocks (a rank that raised locally still takes part, so nobody is left inside a barrier), and a candidate of
    the peer-mapped forms that failed or timed out is torn down on every rank -- error word cleared, peers unmapped --
    so that the next attach starts from zeroed flags and epochs (a half-finished sequence leaves the ranks' exchange
    epochs apart, and the waits compare with >=)."""
    import time
    snap = [t.clone() for t in (engine.params, engine.exp_avg, engine.exp_avg_sq)]

    def restore():
        for dst, src in zip((engine.params, engine.exp_avg, engine.exp_avg_sq), snap):
            dst.copy_(src)
        engine.invalidate_staging()

    def sync():
        if torch.device(engine.device).type == "cuda":
            torch.cuda.synchronize(engine.device)

    def attempt(n):
        """`run_steps(n)` + device sync on this rank; -> error text or None.  Never raises."""
        try:
            run_steps(n)
            sync()
            return None
        except Exception as exc:                                   # noqa: BLE001
            return str(exc)[:300]

    def drop(form):
        """A failed candidate leaves nothing behind (collective: every rank calls it)."""
        if form in ("p2p", "p2p_push") and engine.has_p2p:
            try:
                sync()
            except Exception:                                      # noqa: BLE001
                pass
            engine.p2p_clear_errors()
            engine.p2p_close()

    report, best = {}, None
    for form in forms:
        why = self.set_exchange_form(engine, form)
        if why:
            report[form] = {"skipped": why}
            continue
        restore()
        err = attempt(warm)
        if not self.all_agree(engine, err is None):
            report[form] = {"error": err or "another rank failed during the warm-up steps"}
            drop(form)
            continue
        dist.barrier(group=self.group)
        t0 = time.perf_counter()
        err = attempt(steps)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=engine.device)
        if not self.all_agree(engine, err is None):
            report[form] = {"error": err or "another rank failed during the timed steps"}
            drop(form)
            continue
        dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=self.group)
        ok = self.replicas_identical(engine)
        bad = self.p2p_timeouts(engine) if form.startswith("p2p") else 0
        report[form] = {"us_per_step": float(dt.item()) / steps * 1e6, "replicas_identical": ok, "peer_wait_timeouts": bad}
        if bad or not ok:
            drop(form)
        elif best is None or report[form]["us_per_step"] < report[best]["us_per_step"]:
            best = form
    restore()
    if best is not None:
        why = self.set_exchange_form(engine, best)           # (re-attaches the peers if a later candidate's failure unmapped them)
        if why:
            report[best]["error"] = "could not be selected again: " + why
            best = None
    if engine.has_p2p:
        engine.p2p_clear_errors()                            # nothing a rejected candidate left behind may fail the first epoch
    return best, report


DataParallel.all_agree = _all_agree
DataParallel.p2p_timeouts = _p2p_timeouts
DataParallel.set_exchange_form = _set_exchange_form
DataParallel.replicas_identical = _replicas_identical
DataParallel.autotune_exchange = _autotune_exchange


def _all_reduce_async(self, tensor):
    """SUM all-reduce that returns a Work handle (None for a single process).  With the nccl
    backend the collective is stream-ordered behind the kernels already queued on the current
    stream, and `work.wait()` only makes the current stream wait for it."""
    if not self.collective:
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True)


DataParallel.all_reduce_async = _all_reduce_async


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).
    Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: several ranks on ONE GPU (PVAE_LOCAL_DEVICE=0) need the gloo backend
    # (PVAE_DIST_BACKEND=gloo) -- RCCL refuses two ranks on the same device
    if os.environ.get("PVAE_LOCAL_DEVICE") is not None:
        local = int(os.environ["PVAE_LOCAL_DEVICE"])
    backend = backend or os.environ.get("PVAE_DIST_BACKEND")
    # a launcher that starts more ranks on this node than it has GPUs (torchrun --nproc-per-node 8 on a 1-GPU box):
    # the ranks share the devices round-robin over gloo, as with the test hooks above -- functional, not a measurement
    # (every rank of the node sees the same LOCAL_WORLD_SIZE and device count, so all take the same branch)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev and int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > ndev and os.environ.get("PVAE_LOCAL_DEVICE") is None:
        local = local % ndev
        backend = backend or "gloo"
        os.environ["PVAE_LOCAL_DEVICE"] = str(local)
        os.environ["PVAE_BENCH_SHARED_GPU"] = "1"
    force = os.environ.get("PVAE_DP_ALWAYS_REDUCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local

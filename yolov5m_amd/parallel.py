"""Single-node data parallelism: one process per GPU, gradients summed with RCCL over xGMI.

The reference has no distributed code at all; this is the new component BASELINE.json asks
for. Design (MI355X: 7 xGMI links x ~153 GB/s per GPU, point-to-point, no switch):
  * every rank holds a full replica; parameters and gradients are ONE flat f32 buffer each
    (21 190 557 elements, 84.8 MB), so the exchange is a handful of large all-reduces instead of 243
    small ones -- per-link-bound ring collectives want few, large messages;
  * gradient convention = SUM over ranks of the local-batch gradients (each rank's loss is already
    multiplied by its local batch size, ultralytics_loss.py:120), i.e. exactly the gradient of the
    single-process loss on the concatenated batch up to BatchNorm's per-replica statistics (the
    reference has no SyncBN to match);
  * the exchange is OVERLAPPED with the backward pass (NativeTrainStep._step_overlapped): the backward launch list is
    cut where the finished part of the flat buffer -- a growing suffix: the head's gradients are final first, the
    stem's last -- reaches 50 % and 90 % of the elements (Engine.grad_cuts; most parameters sit in the deep layers, so
    the first cut comes after about a fifth of the backward time). Each segment is its own captured hipGraph; after a
    segment is enqueued, GradAllReduce.launch starts the asynchronous SUM all-reduce of the range it finished (on the
    backend's stream, ordered after the segment) and it runs under the next segment's data / weight gradients; all
    are waited for before the optimizer graph. Three buckets of ~42 / 34 / 8 MB: few, large messages suit rings that are
    bound per xGMI link. `overlap=False` (or a plain callable as grad_hook) gives one all-reduce after the backward pass;
  * the optimizer then runs identically on every rank (no parameter broadcast after step 0). Gradients are SUMMED, and
    the reference's clip_grad_norm_(10) (train.py:118) is applied to that sum: with W ranks the clip engages at 1/W of
    the per-replica gradient norm -- the same as the single-process recipe on the W-times larger batch, whose loss is
    also a sum over images (ultralytics_loss.py:120). Rescale max_norm by W to keep the per-replica threshold instead;
  * tools/dp_parity.py (tests/test_gpu_zz_dp.py, two ranks): exchanged gradient == sum of the single-replica gradients,
    parameters bit-identical across ranks after captured steps, overlapped == plain schedule. No scaling curve has been
    measured by this build (one GPU per box): the driver's SCALE run is the first multi-GPU execution.
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU tensors (tests). `Y5M_DIST_BACKEND=gloo`
forces gloo on GPU tensors too: with LOCAL_RANK folded onto the visible devices this lets a 1-GPU box run the
whole multi-process flow (RCCL itself refuses two ranks on one device).
"""
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the persistent kernels (one workgroup per CU: halo-patch conv, long-K GEMM, fused backward kernels) leave 16 CUs to
        # the collective's kernels when there IS a collective: on one GPU a cap of 224-240 CUs measured the same step time as all
        # 256 (profiles/r03_knob_sweep.txt), so the headroom costs nothing; read once by the library (y5m_persistent_cus)
        os.environ.setdefault("Y5M_PERSIST_CUS", "240")
        backend = backend or os.environ.get("Y5M_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if backend != "nccl":
                local %= torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def make_buckets(numel, boundaries, target_bytes=32 << 20):
    """Split [0, numel) of the flat gradient buffer into buckets in BACKWARD order.

    boundaries: element offsets (ascending) where a bucket may start (= starts of the layer units).
    Returns [(lo, hi)] ordered from the END of the buffer (head: first gradients to be final) to the
    start (stem: last), each about target_bytes of f32."""
    cuts = sorted(set([0, numel] + [b for b in boundaries if 0 < b < numel]))
    target = max(1, target_bytes // 4)
    out = []
    hi = numel
    i = len(cuts) - 2
    while hi > 0:
        lo = cuts[i]
        while i > 0 and hi - lo < target:
            i -= 1
            lo = cuts[i]
        out.append((lo, hi))
        hi = lo
        i -= 1
    return out


class GradAllReduce:
    """SUM all-reduce of a flat gradient buffer, whole (`hook(flat)`) or bucket by bucket (`launch` / `wait`).

    Bucketed form on GPU tensors: every bucket is issued from a dedicated COMMUNICATION stream that first waits for an
    event recorded on the compute stream (everything enqueued so far -- the backward segment that finished the bucket),
    so the compute stream itself never waits for a collective until `wait()`, right before the optimizer. With
    `timing=True` each bucket carries an event pair on the communication stream (ready -> reduced: queueing behind the
    previous bucket included) and `wait()` is bracketed by a pair on the compute stream: that pair's elapsed time is the
    EXPOSED part of the exchange (`stats()`, read by bench.py after a synchronize).
    `force=True` runs the collectives with a single rank too (the 1-GPU RCCL smoke of the overlapped schedule)."""

    def __init__(self, world_size=None, group=None, timing=False, force=False):
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.timing, self.force = timing, force
        self.pending = []
        self._comm = None
        self._evs = []           # per bucket index: (ready, start, end), created once and re-recorded every step
        self._wait_evs = None
        self._buckets = []       # (lo, hi) of the last step's launches, in launch order

    @property
    def active(self):
        return self.world > 1 or (self.force and dist.is_initialized())

    def __call__(self, flat):
        """blocking (stream-ordered on GPU) all-reduce of the whole buffer"""
        if self.active:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def launch(self, flat, lo, hi):
        """asynchronous all-reduce of flat[lo:hi] (call after the producing kernels were enqueued)"""
        if not self.active:
            return
        if not flat.is_cuda:
            self.pending.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        i = len(self.pending)
        if i == 0:
            self._buckets = []
        self._buckets.append((lo, hi))
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=flat.device)
        while len(self._evs) <= i:
            self._evs.append(tuple(torch.cuda.Event(enable_timing=self.timing) for _ in range(3)))
        ready, start, end = self._evs[i]
        main = torch.cuda.current_stream(flat.device)
        ready.record(main)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ready)
            if self.timing:
                start.record(self._comm)
            w = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()            # nccl: the communication stream waits for the backend's stream (no host block); gloo: host wait
            end.record(self._comm)
        self.pending.append(end)

    def wait(self):
        """the current stream waits for every launched bucket"""
        if not self.pending:
            return
        if not isinstance(self.pending[0], torch.cuda.Event):
            for w in self.pending:
                w.wait()
            self.pending = []
            return
        main = torch.cuda.current_stream()
        if self.timing:
            if self._wait_evs is None:
                self._wait_evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._wait_evs[0].record(main)
        for end in self.pending:
            main.wait_event(end)
        if self.timing:
            self._wait_evs[1].record(main)
        self.pending = []

    def stats(self):
        """timing of the LAST step's exchange (call after torch.cuda.synchronize(); needs timing=True): per bucket
        (megabytes, ms from "gradients final" to "reduced" on the communication stream) and the milliseconds the compute
        stream spent waiting for the exchange in wait()."""
        if not self.timing or self._wait_evs is None or not self._buckets:
            return None
        b = [{"MB": round((hi - lo) * 4 / 1e6, 2), "ms": round(self._evs[i][1].elapsed_time(self._evs[i][2]), 3)}
             for i, (lo, hi) in enumerate(self._buckets)]
        return {"buckets": b, "allreduce_exposed_ms": round(self._wait_evs[0].elapsed_time(self._wait_evs[1]), 3)}


def broadcast_parameters(model, src=0, group=None):
    """make every replica start from rank `src`'s parameters and BN buffers"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if getattr(model, "flat_params", None) is not None:
        dist.broadcast(model.flat_params, src=src, group=group)
        dist.broadcast(model._flat_stats, src=src, group=group)
        dist.broadcast(model._nbt, src=src, group=group)       # BatchNorm2d.num_batches_tracked
        model.mark_weights_changed()       # (a collective bumps no torch version counter: a pack_once inference plan must pack again)
    else:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)

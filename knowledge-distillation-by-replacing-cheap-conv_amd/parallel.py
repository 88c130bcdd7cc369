"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in the CPU tests).  The reference has no working multi-GPU path (nn.DataParallel breaks its
trainers); this replaces it with a bucketed all-reduce of the *trainable* gradients only -- the cheap-conv
blocks, 7-9 M fp32 values (~35 MB) -- launched from inside the student's backward as soon as a bucket's last gradient
kernel has been enqueued, on a side stream, so the exchange overlaps the rest of backward.

Payload sizing for xGMI (point-to-point links, ring collectives are per-link bound): ~35 MB total in 2-4 buckets of
>= 8 MB keeps every message far above the latency regime while leaving the last bucket small; BN needs no sync
(eval-mode statistics), and every loss is an element mean, so averaging per-rank gradients over equal
shards reproduces the global-batch gradient.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes=8 << 20, process_group=None):
        """params: trainable parameters in the order their gradients are PRODUCED by backward (reverse of forward)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a 1-rank group normally skips the collective; tests force it to exercise the stream / RCCL plumbing on one GPU
        self.force_collective = False
        self.params = list(params)
        self.buckets = []      # list of dict(flat=tensor, views={param: view}, pending=int, work=None)
        self._where = {}
        cur, cur_bytes = [], 0
        for p in self.params:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._stream = None
        self._set_conv_grid_cap()

    # Workgroups the persistent conv kernels launch while this reducer exchanges buckets from inside backward.  Measured on one
    # MI355X (tools/sidestream_probe.py -> profiles/r05_sidestream.json: the headline mode-A step with a stand-in exchange kernel
    # per bucket on the side stream): a side-stream kernel that becomes runnable when a bucket's last gradient kernel retires
    # starts 13 us later whether the conv grids hold 256, 248, 240 or 224 workgroups -- it is dispatched at the kernel boundary,
    # ahead of the next conv launch's workgroups -- and it runs for the same 0.2 ms under the next conv launch either way, while
    # every 8 CUs withheld from the conv kernels cost 3.3 % of the step (178.97 -> 184.93 -> 186.43 -> 190.99 ms).  So the cap this
    # reducer sets for world > 1 is NONE (one workgroup per CU); KDCC_PERSIST_CUS in the environment still overrides it for an
    # A/B on the 8-GPU node, where the exchange is a multi-step RCCL ring instead of one copy.
    CONV_GRID_CAP = 0

    def _set_conv_grid_cap(self):
        import os
        if self.world > 1 and torch.cuda.is_available() and "KDCC_PERSIST_CUS" not in os.environ:
            from . import _lib
            _lib.check(_lib.lib().kd_conv_set_persist_cus(self.CONV_GRID_CAP), "kd_conv_set_persist_cus")

    def _close(self, plist):
        dev = plist[0].device
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
        views, off = {}, 0
        for p in plist:
            views[p] = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        b = dict(flat=flat, views=views, pending=len(plist), total=len(plist), work=None, event=None)
        for p in plist:
            self._where[p] = b
        self.buckets.append(b)

    # --- engine-facing hooks ------------------------------------------------------------------
    def grad_buffer(self, p):
        """Tensor the wgrad kernel should write p's gradient into (a slice of its bucket)."""
        return self._where[p]["views"][p]

    def grad_ready(self, p):
        """The kernel producing p's gradient has been enqueued on the current stream."""
        b = self._where[p]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1 and not (self.force_collective and dist.is_initialized()):
            return
        flat = b["flat"]
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Make the current stream wait for every bucket, average, and re-arm for the next step."""
        for b in self.buckets:
            if b["pending"] != 0 and b["pending"] != b["total"]:
                raise RuntimeError("GradReducer: a bucket was only partly produced by backward")
            if b["work"] is not None:
                if b["flat"].is_cuda:
                    with torch.cuda.stream(self._stream):
                        b["work"].wait()
                        b["flat"].div_(self.world)
                else:
                    b["work"].wait()
                    b["flat"].div_(self.world)
                b["work"] = None
            b["pending"] = b["total"]
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


def init_distributed(use_cuda=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and, when WORLD_SIZE > 1, creates the
    process group once: backend "nccl" (= RCCL over xGMI) when the run uses the GPUs, "gloo" otherwise.  Idempotent --
    ConfigParser, BaseTrainer and bench.py all call it.  Returns (rank, local_rank, world).

    Rehearsal hooks for a one-GPU box (never set on the node): KDCC_DIST_SHARE_GPU=1 maps every local rank onto the
    devices that exist (local_rank % device_count) and KDCC_DIST_BACKEND=gloo exchanges the device tensors through gloo,
    since RCCL refuses two ranks on one device -- the N > 1 code path of bench.py / the trainers then runs end to end."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available() if use_cuda is None else (bool(use_cuda) and torch.cuda.is_available())
    if cuda:
        if os.environ.get("KDCC_DIST_SHARE_GPU") == "1":
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    if dist.is_initialized():
        return dist.get_rank(), local, dist.get_world_size()
    if world > 1:
        backend = os.environ.get("KDCC_DIST_BACKEND") or ("nccl" if cuda else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def mean_scalar(value):
    """Rank mean of a python number / 0-dim tensor (identity without a process group).  The trainers feed the result to
    everything that steers control flow -- plateau LR scheduler, best-metric monitor, early stop -- so that all ranks
    take the same decision from their different data shards."""
    value = float(value)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()


def barrier():
    """All ranks meet (no-op without a process group)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def broadcast_object(obj, src=0):
    """Rank `src`'s picklable object on every rank (e.g. the run id that names the checkpoint directory)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, device=_comm_device())
    return box[0]


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers (freshly created cheap-conv blocks are drawn
    from each process's own RNG)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if dist.get_backend() == "nccl" and not t.is_cuda:
                continue
            dist.broadcast(t.data, src=src)

"""Process-group bootstrap and the data-parallel gradient reducer (reference util/distributed.py,
and the DistributedDataParallel wrap of util/misc.py:32-34).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm and runs over xGMI inside a
node.  The reducer all-reduces contiguous slices ("buckets") of the flat gradient buffer on a side
stream as soon as every parameter of a bucket has its final gradient (the engine calls
``grad_ready``), so communication overlaps the rest of the hand-written backward; there is no
flatten/unflatten copy and no per-parameter autograd hook.  Buckets are sized for xGMI
(large messages: 7 links x ~153 GB/s per GPU are only reached by multi-MB collectives).
The reference launcher's NCCL_P2P_DISABLE=1 is deliberately NOT carried over (it would disable xGMI).
"""
from __future__ import annotations

import builtins
import contextlib
import datetime
import os
import random
import sys
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .. import engine


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master, log_fn=None):
    """Rank-0-only, time-stamped print that is also appended to a log file (util/distributed.py:13-34)."""
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop('force', False) or (get_rank() % 8 == 0)
        if is_master or force:
            msg = f'[{datetime.datetime.now().time()}] ' + ' '.join(str(a) for a in args)
            builtin_print(msg, **kwargs)
            sys.stdout.flush()
            if log_fn is not None:
                with open(log_fn, 'a') as f:
                    f.write(msg + '\n')
    builtins.print = print


def _cfg_get(node, key, default=None):
    try:
        return getattr(node, key)
    except (AttributeError, KeyError):
        return default


def _seed_everything(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)


def init_distributed_mode(local_rank, args, log_fn=None):
    """util/distributed.py:66-100 with env:// support (torchrun) next to the reference's explicit dist_url.

    A process group is created only when there really are peer processes: under ``torch.distributed.run``
    (WORLD_SIZE > 1 in the environment) or when the launcher spawned one worker per GPU and says so through
    ``env.spawned`` (train.py does that for ``env.ngpu > 1``, as the reference's launcher.py does).  A plain
    ``python train.py`` on a multi-GPU host stays single-process on ``cuda:local_rank`` instead of waiting for
    ranks nobody started.  The RNG streams are seeded on every path (the reference's ``distributed`` is true for any
    GPU run, so its single-GPU runs are seeded too)."""
    ngpus = torch.cuda.device_count()
    env = args.env
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    try:
        spawned = bool(env.spawned)
    except (AttributeError, KeyError):
        spawned = False
    # ranks per node of a spawned launch = env.ngpu (what train.py / the reference's launcher.py spawned), NOT the number of
    # visible devices: env.ngpu = 2 on an 8-GPU host would otherwise wait for six ranks nobody started
    per_node = int(_cfg_get(env, 'ngpu') or ngpus or 1) if spawned else ngpus
    if spawned and per_node > ngpus:
        raise RuntimeError(f'env.ngpu = {per_node} but only {ngpus} device(s) are visible')
    env.distributed = ngpus > 0 and (env_world > 1 or (spawned and env.world_size * max(per_node, 1) > 1))
    if not env.distributed:
        setup_for_distributed(is_master=True, log_fn=log_fn)
        env.world_size, env.rank = 1, 0
        if ngpus > 0:
            torch.cuda.set_device(local_rank)
        if _cfg_get(env, 'seed') is not None:
            _seed_everything(env.seed)
        return
    if 'RANK' in os.environ and env_world > 1:                   # launched by torch.distributed.run
        env.rank, env.world_size = int(os.environ['RANK']), env_world
        local_rank = int(os.environ.get('LOCAL_RANK', local_rank))
        url = 'env://'
    else:
        env.world_size = per_node * env.world_size
        env.rank = env.rank * per_node + local_rank
        url = env.dist_url
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend='nccl', init_method=url, world_size=env.world_size, rank=env.rank,
                            timeout=datetime.timedelta(seconds=int(os.environ.get('DAV_DIST_TIMEOUT_S', '1800'))))
    dist.barrier()
    setup_for_distributed(env.rank == 0, log_fn=log_fn)
    if _cfg_get(env, 'seed') is not None:
        _seed_everything(env.seed + get_rank())   # per-rank stream -> different masks per rank (util/distributed.py:90-94)


@torch.no_grad()
def concat_all_gather(tensor):
    if not is_dist_avail_and_initialized():
        return tensor
    out = [torch.ones_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor, async_op=False)
    return torch.cat(out, dim=0)


class GradReducer:
    """Bucketed, overlapped reduction (average) of FlatParams.flat_g over the data-parallel group.

    Switches for the first multi-GPU runs (all read at construction; README "Data-parallel switches"):
      DAV_DP_ALGO=allreduce|rs_ag   one all-reduce per bucket (default, what DistributedDataParallel issues: util/misc.py:32-34)
                                    or reduce-scatter + all-gather per bucket (the two halves of a ring all-reduce as separate
                                    collectives: same bytes on the wire, lets RCCL pick a different protocol per half)
      DAV_DP_BUCKET_MB / DAV_DP_FIRST_BUCKET_MB   bucket sizes in MB of fp32 gradients (64 / 8) when the caller passes none
      DAV_DP_BF16=1                 NOT the reference's arithmetic (opt-in): a bucket is cast to bf16, summed in bf16 on the wire
                                    (half the bytes: the N = 2 point is bound by ONE xGMI link) and cast back into the fp32
                                    gradient buffer
      DAV_DP_SEGMENTS (util.misc.GraphedStep)   graphs per captured step between which the buckets are reduced (default 5)
    ``skip_collectives`` (attribute; bench.py): run the whole schedule but issue no collective — the step time without
    communication, from which bench.py derives ``comm_ms_exposed``."""

    def __init__(self, flat, bucket_mb: Optional[float] = None, first_bucket_mb: Optional[float] = None, process_group=None):
        if bucket_mb is None:
            bucket_mb = float(os.environ.get('DAV_DP_BUCKET_MB', '64'))
        if first_bucket_mb is None:
            first_bucket_mb = float(os.environ.get('DAV_DP_FIRST_BUCKET_MB', '8'))
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if is_dist_avail_and_initialized() else 1
        self.rank = dist.get_rank(process_group) if is_dist_avail_and_initialized() else 0
        # test hook: run the collective code path even on a 1-rank group (exercises RCCL + streams on a single GPU)
        self.force = os.environ.get('DAV_FORCE_DIST', '0') == '1' and is_dist_avail_and_initialized()
        self.algo = os.environ.get('DAV_DP_ALGO', 'allreduce')
        if self.algo not in ('allreduce', 'rs_ag'):
            raise ValueError(f'DAV_DP_ALGO={self.algo!r}: allreduce or rs_ag')
        self.bf16_wire = os.environ.get('DAV_DP_BF16', '0') == '1'
        self.skip_collectives = False
        self._stage = None                 # bf16 staging buffer of the largest bucket (DAV_DP_BF16=1)
        self.buckets = self._make_buckets(flat, int(first_bucket_mb * 2 ** 20 // 4), int(bucket_mb * 2 ** 20 // 4))
        self._bucket_of = {}
        for bi, (lo, hi, plist) in enumerate(self.buckets):
            for p in plist:
                self._bucket_of[id(p)] = bi
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._streams = [set() for _ in self.buckets]      # streams on which a bucket's gradients were finished
        self.enabled = True
        # DDP semantics: whether a backward reduces its gradients is decided when its FORWARD runs (train.py wraps only the
        # forward in no_sync(); the backward happens later in Trainer.step, outside the context manager)
        self.sync_this_backward = True
        self.comm_stream = torch.cuda.Stream() if flat.flat_g.is_cuda else None
        self.launch_order: List[int] = []
        # ReduceOp.AVG is native in RCCL; probe it once and fall back to SUM + scale if this build rejects it
        self.use_avg = False
        if (self.world > 1 or self.force) and flat.flat_g.is_cuda:
            try:
                probe = torch.ones(8, device=flat.flat_g.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.group)
                torch.cuda.synchronize()
                self.use_avg = bool(abs(float(probe[0]) - 1.0) < 1e-6)
            except Exception:
                self.use_avg = False

    @staticmethod
    def _make_buckets(flat, first_elems, cap_elems):
        buckets, lo, cur, cap = [], 0, [], first_elems
        ends = flat.seg_end.tolist()
        for p, end in zip(flat.params, ends):
            cur.append(p)
            if end - lo >= cap:
                buckets.append((lo, end, cur))
                lo, cur, cap = end, [], cap_elems
        if cur:
            buckets.append((lo, flat.total, cur))
        return buckets

    def begin_backward(self):
        self.sync_this_backward = self.enabled
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._streams = [set() for _ in self.buckets]
        self.launch_order = []

    def grad_ready(self, p):
        """Engine hook: the last gradient kernel of ``p`` has been enqueued on the compute stream."""
        bi = self._bucket_of.get(id(p))
        if bi is None or not self.sync_this_backward:
            return
        if self.comm_stream is not None:
            # the engine finishes gradients on several streams (towers / decoders run on their own): the reduction has
            # to wait for every one of them, not only for the stream of the parameter that happens to report last
            self._streams[bi].add(torch.cuda.current_stream())
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self._launched[bi]:
            self._launch(bi)

    def _reduce_avg(self, view):
        """Average ``view`` (a 1-D slice of a gradient buffer, fp32 or the bf16 staging copy) over the group, in place, on the
        current stream, with the configured algorithm."""
        on_gpu = view.is_cuda
        op = dist.ReduceOp.AVG if (self.use_avg and on_gpu) else dist.ReduceOp.SUM
        n, w = view.numel(), self.world
        chunk = n // w if self.algo == 'rs_ag' else 0
        if chunk > 0:
            # reduce-scatter + all-gather over the first chunk * world elements (rank r owns [r * chunk, (r + 1) * chunk): both
            # collectives in place, NCCL's documented in-place layouts), a plain all-reduce for the < world leftover elements
            main = view[:chunk * w]
            mine = main[self.rank * chunk:(self.rank + 1) * chunk]
            if on_gpu:
                dist.reduce_scatter_tensor(mine, main, op=op, group=self.group)
                dist.all_gather_into_tensor(main, mine, group=self.group)
            else:      # gloo (CPU tests) has no reduce-scatter: one rooted reduce per shard is the same arithmetic
                for r in range(w):
                    dist.reduce(main[r * chunk:(r + 1) * chunk], dst=dist.get_global_rank(self.group, r) if self.group is not None else r,
                                op=dist.ReduceOp.SUM, group=self.group)
                parts = [torch.empty_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine.clone(), group=self.group)
                for r in range(w):
                    main[r * chunk:(r + 1) * chunk].copy_(parts[r])
            if n > chunk * w:
                dist.all_reduce(view[chunk * w:], op=op, group=self.group)
        else:
            dist.all_reduce(view, op=op, group=self.group)
        if op == dist.ReduceOp.SUM and w > 1:
            view.mul_(1.0 / w)

    def _reduce_bucket(self, view):
        if not self.bf16_wire:
            self._reduce_avg(view)
            return
        # opt-in, NOT the reference's arithmetic: bf16 on the wire (fp32 -> bf16, reduce, bf16 -> fp32 back into the buffer)
        if self._stage is None or self._stage.numel() < view.numel() or self._stage.device != view.device:
            self._stage = torch.empty(max(hi - lo for lo, hi, _ in self.buckets), dtype=torch.bfloat16, device=view.device)
        st = self._stage[:view.numel()]
        st.copy_(view)
        self._reduce_avg(st)
        view.copy_(st)

    def _launch(self, bi):
        self._launched[bi] = True
        self.launch_order.append(bi)
        if (self.world == 1 and not self.force) or self.skip_collectives:
            return
        lo, hi, _ = self.buckets[bi]
        view = self.flat.flat_g[lo:hi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for st in self._streams[bi]:
                self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                self._reduce_bucket(view)
        else:                                     # gloo (CPU tests): no AVG op
            self._reduce_bucket(view)

    def launch_buckets(self, bucket_ids):
        """All-reduce the given buckets now (their gradients are complete on the current stream): used by the
        segmented hipGraph step, which learns at capture time which buckets each graph segment completes."""
        for bi in bucket_ids:
            if not self._launched[bi]:
                self._launch(bi)

    def finish(self):
        """Launch whatever is still pending (in bucket order) and make the compute stream wait for the reductions."""
        if not self.sync_this_backward:
            return
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        if self.comm_stream is not None and (self.world > 1 or self.force):
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def reduce_all_now(self):
        """Non-overlapped variant: all-reduce every bucket after the backward finished."""
        self.begin_backward()
        self.finish()


class DataParallel(torch.nn.Module):
    """Stand-in for ``DistributedDataParallel(model)`` (util/misc.py:34): same forward, ``no_sync()`` for gradient
    accumulation (util/misc.py:144-148), parameters broadcast from rank 0 at construction."""

    def __init__(self, module: torch.nn.Module, flat, bucket_mb: Optional[float] = None, first_bucket_mb: Optional[float] = None,
                 process_group=None):
        super().__init__()
        self.module = module
        self.reducer = GradReducer(flat, bucket_mb=bucket_mb, first_bucket_mb=first_bucket_mb, process_group=process_group)
        if self.reducer.world > 1 or self.reducer.force:
            dist.broadcast(flat.flat_p, src=0, group=process_group)           # C2 in SURVEY.md section 2c
            for b in module.buffers():
                dist.broadcast(b, src=0, group=process_group)
            for p in module.parameters():
                if not p.requires_grad:
                    dist.broadcast(p.data, src=0, group=process_group)
        engine.set_grad_ready_hook(self.reducer.grad_ready)

    def forward(self, *args, **kwargs):
        self.reducer.begin_backward()
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        old = self.reducer.enabled
        self.reducer.enabled = False
        try:
            yield
        finally:
            self.reducer.enabled = old

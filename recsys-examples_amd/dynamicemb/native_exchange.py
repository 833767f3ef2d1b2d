"""The exchanges of a row-wise sharded step through the library's own RCCL calls (csrc/exchange.hip): one C call per stage.

`RowWiseShardedLookup` (sharded.py) routes GPU batches of the exact exchange here; CPU tensors (the gloo tests), the
fixed-capacity exchange and `collapse_batch` keep the c10d call sequence of `RwSparseFeaturesDist` (input_dist.py), which is
also what `MI355_NATIVE_EXCHANGE=0` selects.  Same buffers, same splits, same order of operations -- only who issues the
collective differs: the tests compare the two paths bit for bit (tests/test_sharded_gpu.py).

Reference: corelib/dynamicemb/dynamicemb/input_dist.py:199-285 (RwSparseFeaturesDist + TorchRec KJTAllToAll),
planner/rw_sharding.py:85-158, 191-261 (the output dists).
"""
from __future__ import annotations

import ctypes
import logging
import math
import os
import time
from typing import Optional

import torch
import torch.distributed as dist

from .input_dist import ShardedKeys


def native_exchange_wanted(pg, t: torch.Tensor) -> bool:
    if not t.is_cuda or os.environ.get("MI355_NATIVE_EXCHANGE", "1") == "0":
        return False
    try:
        return str(dist.get_backend(pg)).lower() == "nccl"
    except Exception:
        return False


_log = logging.getLogger("dynamicemb.native_exchange")


def _rccl_candidates():
    """the librccl.so this process already maps (torch.distributed's "nccl" backend IS RCCL), then the usual places"""
    paths = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    q = line.split()[-1]
                    if q not in paths:
                        paths.append(q)
    except OSError:
        pass
    paths.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    paths += ["librccl.so", "librccl.so.1"]
    return paths


def all_ranks_ok(pg, ok: bool, device) -> bool:
    """ONE decision for all ranks: a rank that fell back alone would leave the others inside a collective"""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
    return bool(flag.item())


def bounded_wait(ready, seconds: Optional[float] = None) -> bool:
    """polls `ready()` (an event query) for at most `seconds`: the hang guard of the self-check"""
    if seconds is None:
        seconds = 30.0
    end = time.monotonic() + seconds
    while not ready():
        if time.monotonic() > end:
            return False
        time.sleep(0.0005)
    return True


class NativeExchange:
    """Two RCCL communicators over the ranks of `pg` (input dist / output dist), created from unique ids that rank 0 draws and
    the existing process group broadcasts.

    Construction never leaves a rank alone: every step that can fail on one rank only (binding librccl.so, drawing the ids,
    ncclCommInitRank) ends in a MIN all-reduce of an ok flag over `pg`; `create()` returns None on EVERY rank if any rank
    failed, and the caller keeps the c10d call sequence (input_dist.py)."""

    @classmethod
    def create(cls, pg, device) -> Optional["NativeExchange"]:
        x = cls.__new__(cls)
        x._h = None
        x.dead = False
        why = None
        try:
            x._bind(pg)
            raw = x._draw_ids()
        except Exception as e:       # noqa: BLE001 -- whatever went wrong, the step continues on the c10d sequence
            why, raw = e, None
        if not all_ranks_ok(pg, why is None, device):
            _log.warning("in-library RCCL exchange unavailable (%s): using the c10d sequence", why or "another rank failed")
            return None
        try:
            x._init_comms(pg, device, raw)
        except Exception as e:       # noqa: BLE001
            why = e
        if not all_ranks_ok(pg, why is None, device):
            _log.warning("in-library RCCL exchange: communicator creation failed (%s): using the c10d sequence",
                         why or "another rank failed")
            x.abort()
            return None
        return x

    def __init__(self, pg, device):
        """raises on failure (single-process use and tests); `create()` is the form that agrees across ranks"""
        self._h = None
        self.dead = False
        self._bind(pg)
        self._init_comms(pg, device, self._draw_ids())

    def _bind(self, pg):
        from mi355_native import NativeError, lib

        self.pg = pg
        self.W = dist.get_world_size(pg)
        self.rank = dist.get_rank(pg)
        self._lib = lib()
        errs = []
        for path in _rccl_candidates():
            if self._lib.mi355_rw_load_rccl(path.encode()) == 0:
                return
            msg = self._lib.mi355_last_error()
            errs.append(f"{path}: {msg.decode() if msg else '?'}")
        raise NativeError("librccl.so could not be bound: " + "; ".join(errs))

    def _draw_ids(self):
        from mi355_native import check

        raw = (ctypes.c_uint8 * 256)()
        if self.rank == 0:
            check(self._lib.mi355_rw_unique_id(ctypes.addressof(raw), 128), "rw_unique_id")
            check(self._lib.mi355_rw_unique_id(ctypes.addressof(raw) + 128, 128), "rw_unique_id")
        return raw

    def _init_comms(self, pg, device, raw):
        from mi355_native import check

        ids = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone().to(device)
        dist.broadcast(ids, src=dist.get_global_rank(pg, 0) if pg is not dist.group.WORLD else 0, group=pg)
        host = (ctypes.c_uint8 * 256).from_buffer_copy(bytes(ids.cpu().numpy().tobytes()))
        torch.cuda.synchronize(device)
        h = ctypes.c_void_p()
        check(self._lib.mi355_rw_create(ctypes.addressof(host), ctypes.addressof(host) + 128, self.W, self.rank,
                                        ctypes.byref(h)), "rw_create")
        self._h = h
        self._ss = (ctypes.c_int64 * self.W)()
        self._rs = (ctypes.c_int64 * self.W)()
        self._tot = (ctypes.c_int64 * 2)()
        self._ticket = ctypes.c_int()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        self.dead = True
        if h is not None:
            try:
                self._lib.mi355_rw_destroy(h)
            except Exception:
                pass

    def abort(self):
        """the hang guard: ncclCommAbort on both communicators (a collective kernel waiting for a peer that never came
        returns); the handle is gone afterwards"""
        h, self._h = getattr(self, "_h", None), None
        self.dead = True
        if h is not None:
            try:
                self._lib.mi355_rw_abort(h)
            except Exception:
                pass

    def __del__(self):
        # (communicators are left to the process teardown: destroying them is a collective-ish call that must not run from a
        #  garbage collector on one rank only)
        pass

    # ------------------------------------------------------------------------------------------------ input dist
    def input_begin(self, F: int, offsets, values, block_sizes, dist_codes, sequence: bool, side: Optional[torch.cuda.Stream]):
        """bucketize -> lengths all-to-all -> received offsets -> per-peer key counts on their way to pinned memory; on the
        stream `side` (behind the current stream's work) or, side = None, on the current stream"""
        from mi355_native import check, ptr, stream

        W = self.W
        FB = offsets.numel() - 1
        if FB % F:
            raise ValueError(f"{FB} bags do not divide into {F} features")
        if values.element_size() != 8:
            raise ValueError("the in-library exchange moves 8-byte keys")
        B = FB // F
        n = values.numel()
        nl = W * FB
        o_new_off = nl
        o_new_val = 2 * nl + 1
        o_perm = o_new_val + n
        o_recv_len = o_perm + (n if sequence else 0)
        o_recv_off = o_recv_len + nl
        buf = torch.empty(o_recv_off + nl + 1, dtype=torch.int64, device=values.device)
        cur = stream()
        st = cur
        if side is not None:
            buf.record_stream(side)       # owned by the caller's stream, written on the exchange stream
            st = ctypes.c_void_p(side.cuda_stream)
        base = buf.data_ptr()
        check(self._lib.mi355_rw_input_begin(self._h, F, B, ptr(offsets), ptr(values), ptr(block_sizes), ptr(dist_codes),
                                             base, base + 8 * o_new_off, base + 8 * o_new_val,
                                             base + 8 * o_perm if sequence else None, base + 8 * o_recv_len,
                                             base + 8 * o_recv_off, cur, st, ctypes.byref(self._ticket)), "rw_input_begin")
        return (buf, self._ticket.value, F, B, n, nl, o_new_val, o_perm if sequence else -1, o_recv_len, o_recv_off, side,
                values.dtype)

    def counts_ready(self, state) -> bool:
        """True when input_finish(state) would not wait for the key counts"""
        return bool(self._lib.mi355_rw_input_counts_ready(self._h, state[1]))

    def input_finish(self, state) -> ShardedKeys:
        """the host reads the key counts, the keys travel; the consumer orders itself behind them with wait_keys()"""
        from mi355_native import check, stream

        buf, ticket, F, B, n, nl, o_new_val, o_perm, o_recv_len, o_recv_off, side, kdt = state
        W = self.W
        check(self._lib.mi355_rw_input_counts(self._h, ticket, self._ss, self._rs, self._tot), "rw_input_counts")
        n_send, n_recv = self._tot[0], self._tot[1]
        send_splits, recv_splits = list(self._ss), list(self._rs)
        recat = W > 1 and F > 1
        rnum = n_recv * 2 + 2 * nl + 1 if recat else n_recv
        st = stream()
        if side is not None:
            # allocated ON the exchange stream: a block of the caller's stream may have been freed by a tensor whose kernels
            # are still queued there (nothing orders `side` behind them -- the fork event exists only in input_begin), and
            # the key exchange would write into it early.  The consumer (the caller's stream, behind wait_keys) is recorded.
            from mi355_native import current_torch_stream, on_stream

            with on_stream(side):
                rbuf = torch.empty(rnum, dtype=torch.int64, device=buf.device)
            rbuf.record_stream(current_torch_stream())
            st = ctypes.c_void_p(side.cuda_stream)
        else:
            rbuf = torch.empty(rnum, dtype=torch.int64, device=buf.device)
        base, rbase = buf.data_ptr(), rbuf.data_ptr()
        check(self._lib.mi355_rw_input_keys(self._h, ticket, F, B, base + 8 * o_new_val, rbase, base + 8 * o_recv_len,
                                            base + 8 * o_recv_off,
                                            rbase + 8 * (2 * n_recv) if recat else None,
                                            rbase + 8 * (2 * n_recv + nl) if recat else None,
                                            rbase + 8 * n_recv if recat else None, st, st), "rw_input_keys")
        recv_offsets = buf[o_recv_off:o_recv_off + nl + 1]
        if recat:
            lengths = rbuf[2 * n_recv:2 * n_recv + nl]
            offs = rbuf[2 * n_recv + nl:2 * n_recv + 2 * nl + 1]
            vals = rbuf[n_recv:2 * n_recv]
        else:
            lengths = buf[o_recv_len:o_recv_len + nl]
            offs = recv_offsets
            vals = rbuf[:n_recv] if rbuf.numel() != n_recv else rbuf
        if kdt != torch.int64:
            vals = vals.view(kdt)
        perm = buf[o_perm:o_perm + n_send] if o_perm >= 0 else None
        sk = ShardedKeys(lengths, offs, vals, recv_offsets, send_splits, recv_splits, perm, B, F)
        sk._ticket = ticket if side is not None else None
        sk._last_ticket = ticket
        return sk

    def keys_ready(self, ticket: int) -> bool:
        return bool(self._lib.mi355_rw_keys_ready(self._h, ticket))

    def wait_keys(self, ticket: int) -> None:
        from mi355_native import check, stream

        check(self._lib.mi355_rw_wait_keys(self._h, ticket, stream()), "rw_wait_keys")

    # ------------------------------------------------------------------------------------------------ output dists
    def output_pooled(self, send: torch.Tensor, out_dtype) -> torch.Tensor:
        """send [W*B, total_D] partial sums in the wire type -> [B, total_D] sums of the W shards in out_dtype"""
        from mi355_native import check, dt, ptr, stream

        B, D = send.size(0) // self.W, send.size(1)
        recv = torch.empty_like(send)
        out = torch.empty(B, D, dtype=out_dtype, device=send.device)
        check(self._lib.mi355_rw_output_pooled(self._h, ptr(send), ptr(recv), B * D, dt(send.dtype), ptr(out), dt(out_dtype),
                                               stream()), "rw_output_pooled")
        return out

    def allgather(self, t: torch.Tensor) -> torch.Tensor:
        from mi355_native import check, ptr, stream

        if self.W == 1:
            return t          # one rank: the gathered tensor IS the rank's block (the consumer only reads it)
        out = torch.empty((self.W * t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        check(self._lib.mi355_rw_allgather(self._h, ptr(t), ptr(out), t.numel() * t.element_size(), stream()), "rw_allgather")
        return out

    def alltoallv_rows(self, send: torch.Tensor, send_counts, recv_counts) -> torch.Tensor:
        """rows [sum(send_counts), D] -> [sum(recv_counts), D]; counts per peer in rows (python lists)"""
        from mi355_native import check, ptr, stream

        W = self.W
        sc = (ctypes.c_int64 * W)(*send_counts)
        rc = (ctypes.c_int64 * W)(*recv_counts)
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        eb = send.element_size() * math.prod(send.shape[1:])
        check(self._lib.mi355_rw_alltoallv(self._h, ptr(send), sc, ptr(out), rc, eb, stream()), "rw_alltoallv")
        return out

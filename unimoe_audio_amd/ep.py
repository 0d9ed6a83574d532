"""Expert-parallel exchange for the DCMoE layer (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI).

Replaces the reference's AudioMOELayer exchange (reference utils/UniMoE_Audio_core.py:455-488): there, every rank pads
each expert's rows to the GLOBAL maximum capacity (an extra scalar MAX all-reduce, core.py:455-457) and ships
[E, C, D] through DeepSpeed's `_AllToAll` twice.  Here:
  * rank r owns routed experts [r*E_loc, (r+1)*E_loc)  (core.py:505);
  * the per-(destination expert) capacity is fixed at S_local rows -- every local row can pick an expert at most once --
    so no collective is needed to agree on a capacity and the buffers are static (hipGraph friendly);
  * counts travel in the same exchange pattern (one small all-to-all), rows in one all-to-all each way;
  * the combine sums experts in ascending expert order on the token's owner, exactly like the single-GPU path.
The index arithmetic below is backend-agnostic torch code (it runs under gloo on CPU in tests/test_ep_gloo.py with
the oracle as the expert function); on a GPU the expert function is the HIP grouped GEMM.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


class UmoeEpComm:
    """RCCL communicator behind the C-ABI (umoe_ep_comm_create / umoe_ep_all_to_all): the exchange is enqueued on the
    caller's HIP stream by the library itself (graph-capturable), instead of going through torch.distributed.  The 128-byte
    unique id travels over the given torch.distributed group (any backend) once, at construction."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        self.rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
        self.size = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_char * 128)()
            L.check(L.lib().umoe_ep_unique_id(C.cast(buf, C.c_void_p)), "umoe_ep_unique_id")
            uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
        if self.size > 1:
            t = uid.to(device) if dist.get_backend(group) == "nccl" else uid
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = t.cpu()
        raw = (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes()))
        self.h = C.c_void_p()
        L.check(L.lib().umoe_ep_comm_create(C.cast(raw, C.c_void_p), self.rank, self.size, C.byref(self.h)), "umoe_ep_comm_create")

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        assert inp.is_cuda and out.is_cuda and inp.is_contiguous() and out.is_contiguous() and inp.numel() == out.numel()
        per = inp.numel() * inp.element_size() // self.size
        self._L.check(self._L.lib().umoe_ep_all_to_all(self.h, inp.data_ptr(), out.data_ptr(), per, self.size,
                                                       torch.cuda.current_stream().cuda_stream), "umoe_ep_all_to_all")
        return out

    def close(self):
        if getattr(self, "h", None):
            self._L.lib().umoe_ep_comm_destroy(self.h)
            self.h = None


def _a2a(out: torch.Tensor, inp: torch.Tensor, group):
    if isinstance(group, UmoeEpComm):
        return group.all_to_all(out, inp) if group.size > 1 else out.copy_(inp)
    if group is None or dist.get_world_size(group) == 1:
        out.copy_(inp)          # the reference's single-process behaviour: identity (utils.py:332-335)
    elif inp.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves host memory only: device slabs are staged through the host (rehearsals of several ranks on one GPU box;
        # a real job runs the "nccl" = RCCL backend, where the device tensors go straight into the collective)
        h_in, h_out = inp.cpu(), torch.empty(inp.shape, dtype=inp.dtype)
        dist.all_to_all_single(h_out, h_in, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, group=group)
    return out


class _AllToAll(torch.autograd.Function):
    """The exchange as an autograd node (DeepSpeed's `_AllToAll`, deepspeed 0.15.1 moe/sharded_moe.py, used at core.py:467,480):
    forward = the all-to-all of equal slabs, backward = the SAME all-to-all applied to the gradient (the exchange is a permutation of
    slabs across ranks and its own transpose)."""

    @staticmethod
    def forward(ctx, group, inp):
        ctx.group = group
        inp = inp.contiguous()
        return _a2a(torch.empty_like(inp), inp, group)

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        return None, _a2a(torch.empty_like(grad), grad, ctx.group)


def ep_pack(h: torch.Tensor, counts: torch.Tensor, offsets: torch.Tensor, slot_token: torch.Tensor, n_real: int,
            ep_size: int):
    """Builds the send buffers: send [ep(dst), S(pos), E_loc, D] (rows of expert e compacted at positions < count[e]) and
    send_cnt [ep(dst), E_loc]."""
    S, D = h.shape
    E_loc = n_real // ep_size
    pos = torch.arange(S, device=h.device)
    cnt = counts[:n_real].long()
    valid = pos[None, :] < cnt[:, None]                                        # [n_real, S]
    src_slot = (offsets[:n_real].long()[:, None] + pos[None, :]).clamp(max=max(slot_token.numel() - 1, 0))
    rows = slot_token.long()[src_slot]                                          # token feeding (expert e, position pos)
    send = h[rows.reshape(-1)].reshape(n_real, S, D) * valid[..., None].to(h.dtype)
    send = send.reshape(ep_size, E_loc, S, D).permute(0, 2, 1, 3).contiguous()  # [ep(dst), S, E_loc, D]
    send_cnt = counts[:n_real].reshape(ep_size, E_loc).contiguous()
    return send, send_cnt


def ep_dispatch(h: torch.Tensor, counts: torch.Tensor, offsets: torch.Tensor, slot_token: torch.Tensor, n_real: int,
                ep_size: int, group=None):
    """h [S, D]; counts/offsets/slot_token = local ragged dispatch tables (umoe_dispatch_build).
    Returns recv [ep(src), S(pos), E_loc, D] and recv_cnt [ep(src), E_loc]: the rows rank `src` routed to MY experts,
    compacted per expert (positions >= count are zero rows).  First all-to-all of the reference (core.py:467)."""
    send, send_cnt = ep_pack(h, counts, offsets, slot_token, n_real, ep_size)
    recv = _AllToAll.apply(group, send)          # (an autograd node: the training path differentiates through the exchange)
    recv_cnt = torch.empty_like(send_cnt)
    _a2a(recv_cnt, send_cnt, group)
    return recv, recv_cnt


def ep_recv_mask(recv_cnt: torch.Tensor, S: int) -> torch.Tensor:
    """[ep*S "tokens" (src, pos)][E_loc] 0/1 mask of the received rows (input of the ragged dispatch on the receiver)."""
    ep, E_loc = recv_cnt.shape
    pos = torch.arange(S, device=recv_cnt.device)
    m = pos[None, :, None] < recv_cnt.long()[:, None, :]                        # [ep, S, E_loc]
    return m.reshape(ep * S, E_loc).to(torch.int32).contiguous()


def ep_return(y: torch.Tensor, group=None) -> torch.Tensor:
    """y [ep(src), S, E_loc, D] expert outputs for the rows received -> [ep(owner), S, E_loc, D] on the token owner
    (second all-to-all, core.py:480)."""
    return _AllToAll.apply(group, y)


def ep_slot_of(slot_of: torch.Tensor, offsets: torch.Tensor, S: int, ep_size: int) -> torch.Tensor:
    """Local slot_of [S, n_real] (slot = offsets[e] + pos) -> row index into the returned [ep*S*E_loc, D] buffer
    (row = (owner*S + pos)*E_loc + e_loc), -1 where the token is not routed to e."""
    n_real = slot_of.shape[1]
    E_loc = n_real // ep_size
    e = torch.arange(n_real, device=slot_of.device)[None, :]
    pos = slot_of.long() - offsets[:n_real].long()[None, :]
    row = ((e // E_loc) * S + pos) * E_loc + (e % E_loc)
    return torch.where(slot_of >= 0, row, torch.full_like(pos, -1)).to(torch.int32).contiguous()


def ep_moe(h: torch.Tensor, disp: dict, n_real: int, ep_size: int, group,
           expert_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor]):
    """Full exchange around `expert_fn(recv [ep, S, E_loc, D], recv_cnt [ep, E_loc]) -> y [ep, S, E_loc, D]`.
    Returns (y_back [ep*S*E_loc, D], slot_of_ep [S, n_real]) ready for the combine."""
    S = h.shape[0]
    recv, recv_cnt = ep_dispatch(h, disp["counts"], disp["offsets"], disp["slot_token"], n_real, ep_size, group)
    y = expert_fn(recv, recv_cnt)
    back = ep_return(y, group)
    return back.reshape(-1, back.shape[-1]), ep_slot_of(disp["slot_of"], disp["offsets"], S, ep_size)


def ep_combine(y_back: torch.Tensor, slot_of_ep: torch.Tensor, moe_w: torch.Tensor) -> torch.Tensor:
    """sum_e moe_w[s, e] * y_back[slot_of_ep[s, e]] in ascending expert order with fp32 accumulation and one rounding (the
    single-GPU combine, core.py:488 + :342), as differentiable torch ops: the training path of an expert-parallel block."""
    S, n_real = slot_of_ep.shape
    acc = torch.zeros((S, y_back.shape[-1]), dtype=torch.float32, device=y_back.device)
    for e in range(n_real):
        so = slot_of_ep[:, e].long()
        sel = (so >= 0).to(torch.float32)[:, None]
        acc = acc + sel * moe_w[:, e:e + 1].float() * y_back[so.clamp(min=0)].float()
    return acc.to(y_back.dtype)


# ----------------------------------------------------------------------------------------------------------------------
# RAGGED exchange (reference core.py:455-488 without its capacity padding).  The padded form above ships ep x S x E_loc rows each way
# whatever the routing is (204 MB per direction and layer at the training shape, more than the reference's own max-capacity padding);
# here a rank sends every destination exactly the slot rows of that destination's experts -- they are CONTIGUOUS in the local ragged
# dispatch order (slots are expert-major, experts are rank-major) -- after a small fixed-size header exchange that carries the row
# counts: per destination {segment rows, count of each of its experts, start of each expert inside the segment}.  The returned rows
# land in the owner's own slot order, so the combine (and, in training, the whole backward) runs on the owner exactly as at ep_size 1.
class RaggedPlan:
    """Index tables of one ragged exchange (host-computed from the headers; the device tensors live on `device`).
    in_splits / out_splits: rows sent to / received from every rank (segments, alignment padding included).
    rows2 [n2]: row of the RECEIVED buffer behind every valid received row, in (local expert, source rank, position) order;
    pos2  [n2]: where that row sits in the per-expert blocks (block q at offsets2[q], `align`-aligned, counts2[q] rows);
    list2 [cap2]: gather list of the per-expert blocks into the received buffer (padding rows point at row 0)."""

    def __init__(self, in_splits, out_splits, rows2, pos2, list2, counts2, offsets2, cap2):
        self.in_splits, self.out_splits = in_splits, out_splits
        self.n_send, self.n_recv = int(sum(in_splits)), int(sum(out_splits))
        self.rows2, self.pos2, self.list2, self.counts2, self.offsets2, self.cap2 = rows2, pos2, list2, counts2, offsets2, int(cap2)


def ep_ragged_plan(offsets: torch.Tensor, counts: torch.Tensor, n_real: int, ep_size: int, group, align: int = 1, device=None) -> RaggedPlan:
    """offsets [>= n_real + 1] / counts [>= n_real]: the local ragged dispatch (slot = offsets[e] + position; offsets[n_real] = all slots).
    One host read of the local tables + one fixed-size header all-to-all; everything else is host arithmetic on a few integers."""
    import numpy as np
    dev = offsets.device if device is None else device
    E_loc = n_real // ep_size
    offs = offsets[: n_real + 1].detach().cpu().numpy().astype(np.int64)
    cnts = counts[:n_real].detach().cpu().numpy().astype(np.int64)
    hdr = np.zeros((ep_size, 1 + 2 * E_loc), dtype=np.int64)
    for d in range(ep_size):
        b = offs[d * E_loc]
        hdr[d, 0] = offs[(d + 1) * E_loc] - b
        hdr[d, 1: 1 + E_loc] = cnts[d * E_loc: (d + 1) * E_loc]
        hdr[d, 1 + E_loc:] = offs[d * E_loc: (d + 1) * E_loc] - b
    rh = _hdr_exchange(torch.from_numpy(hdr), group, ep_size, dev).numpy()
    in_splits = [int(v) for v in hdr[:, 0]]
    out_splits = [int(v) for v in rh[:, 0]]
    base = np.concatenate([[0], np.cumsum(rh[:, 0])])
    rows2, pos2, counts2, offsets2 = [], [], [], [0]
    run = 0
    for q in range(E_loc):
        start = run
        for src in range(ep_size):
            c = int(rh[src, 1 + q])
            r0 = int(base[src] + rh[src, 1 + E_loc + q])
            rows2.append(np.arange(r0, r0 + c, dtype=np.int64))
            pos2.append(np.arange(run, run + c, dtype=np.int64))
            run += c
        counts2.append(run - start)
        run = (run + align - 1) // align * align
        offsets2.append(run)
    cap2 = max(run, align)
    rows2 = np.concatenate(rows2) if rows2 else np.zeros(0, np.int64)
    pos2 = np.concatenate(pos2) if pos2 else np.zeros(0, np.int64)
    list2 = np.zeros(cap2, dtype=np.int32)
    list2[pos2] = rows2
    # (offsets2[q] = start of block q; the last entry = end of the last block, like umoe_dispatch_build_aligned)
    offsets2 = [offsets2[q] if q == 0 else offsets2[q] for q in range(E_loc)] + [offsets2[E_loc]]
    t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev)
    return RaggedPlan(in_splits, out_splits, t(rows2, torch.int64), t(pos2, torch.int64), t(list2, torch.int32), t(counts2, torch.int32),
                      t(offsets2, torch.int32), cap2)


def _hdr_exchange(send: torch.Tensor, group, ep_size: int, dev) -> torch.Tensor:
    """The fixed-size header all-to-all ([ep, 1 + 2 E_loc] int64 host tensor: row d goes to rank d).  A module-level function so that
    single-process rehearsals can substitute a rendezvous for the collective (tests/test_gpu_ops.py)."""
    if group is None or ep_size == 1 or dist.get_world_size(group) == 1:
        return send.clone()
    if dist.get_backend(group) == "nccl":
        r = torch.empty_like(send, device=dev)
        dist.all_to_all_single(r, send.to(dev), group=group)
        return r.cpu()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv


def ep_exchange_rows(buf: torch.Tensor, in_splits, out_splits, group) -> torch.Tensor:
    """Ragged all-to-all of ROWS: buf [sum(in_splits), D] (segment d goes to rank d) -> [sum(out_splits), D] (segment s came from rank s)."""
    n_in, n_out = int(sum(in_splits)), int(sum(out_splits))
    assert buf.shape[0] >= n_in
    buf = buf[:n_in].contiguous()
    return _rows_exchange(buf, list(in_splits), list(out_splits), n_out, group)


def _rows_exchange(buf: torch.Tensor, in_splits, out_splits, n_out: int, group) -> torch.Tensor:
    if group is None or dist.get_world_size(group) == 1:
        return buf.clone()
    out = torch.empty((n_out,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    if buf.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal of several ranks on one GPU box: staged through the host
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, buf.cpu(), output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, buf, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=group)
    return out


def ep_moe_ragged(h: torch.Tensor, disp: dict, n_real: int, ep_size: int, group,
                  expert_fn: Callable[[torch.Tensor, RaggedPlan], torch.Tensor], align: int = 1):
    """Full ragged exchange around `expert_fn(recv [n_recv, D], plan) -> y2 [cap2, D]` (rows of local expert q at plan.offsets2[q], in
    (source rank, position) order: gather them from `recv` with plan.list2).  Returns y_slots [n_slots, D] in the OWNER's slot order:
    combine it with the local slot_of table exactly as at ep_size 1."""
    plan = ep_ragged_plan(disp["offsets"], disp["counts"], n_real, ep_size, group, align=align, device=h.device)
    st = disp["slot_token"].long()
    if st.numel() < plan.n_send:
        st = torch.nn.functional.pad(st, (0, plan.n_send - st.numel()))
    xs = h[st[: plan.n_send].clamp(0, max(h.shape[0] - 1, 0))]                  # slot order; alignment padding rows are never used
    recv = ep_exchange_rows(xs, plan.in_splits, plan.out_splits, group)
    y2 = expert_fn(recv, plan)
    ret = torch.zeros((plan.n_recv, h.shape[1]), dtype=h.dtype, device=h.device)
    ret[plan.rows2] = y2[plan.pos2]
    return ep_exchange_rows(ret, plan.out_splits, plan.in_splits, group), plan


# ----------------------------------------------------------------------------------------------------------------------
def dense_ep_moe(h: torch.Tensor, expert_fn: Callable[[int, torch.Tensor], torch.Tensor], rank: int, size: int, n_real: int,
                 group=None) -> torch.Tensor:
    """The DENSE exchange of the expert-parallel decode engine (csrc/umoe_engine.hip run_moe_ep), restated with
    torch.distributed collectives for any backend -- the layout contract the engine's slabs follow:
      gather   xg[t] = rows of rank t                                  (every rank's `rows` normalised rows visit every expert)
      local    y_loc[t][x] = expert (rank*E_loc + x) applied to xg[t]  (E_loc = n_real / size local experts, core.py:505)
      return   rank t receives y_loc[t] of every rank src -> dense layout row (src*E_loc + x)*rows + s = (global expert, row)
    h [rows, D]; expert_fn(global expert id, x [n, D]) -> [n, D].  Returns y [n_real, rows, D]; the caller combines by its own
    routing mask in ascending expert order, exactly like ep_size 1."""
    rows, D = h.shape
    E_loc = n_real // size
    if size == 1:
        xg = h[None]
    else:
        parts = [torch.empty_like(h) for _ in range(size)]
        dist.all_gather(parts, h.contiguous(), group=group)
        xg = torch.stack(parts, 0)
    y_loc = torch.empty((size, E_loc, rows, D), dtype=h.dtype, device=h.device)
    for x in range(E_loc):
        y_loc[:, x] = expert_fn(rank * E_loc + x, xg.reshape(size * rows, D)).reshape(size, rows, D)
    y_ret = torch.empty_like(y_loc)
    _a2a(y_ret, y_loc.contiguous(), group if size > 1 else None)
    return y_ret.reshape(n_real, rows, D)


# ----------------------------------------------------------------------------------------------------------------------
# Expert parallel DECODE: the exchange lives inside the engine's captured step graph (include/umoe.h "Peer exchange").
class EpLink:
    """How the decode engines of an expert-parallel job reach each other.

    mode "peer"     one process per GPU: every rank exports its engine's exchange region as a HIP IPC handle, the handles
                    travel over `group` (any torch.distributed backend; objects, once) and every rank maps its peers'
                    regions -- afterwards the step graph stores straight into the peers' memory over xGMI.
         "rccl"     the same exchange as ncclAllGather + grouped ncclSend/ncclRecv on a communicator of the C-ABI
                    (UmoeEpComm), enqueued by the library on the engine's stream.
         "loopback" single-GPU emulation of ONE rank of a `size`-rank job (timing only, see umoe_engine_ep_connect).
    `EpLink.local_mesh(engines)` connects several engines of ONE process (virtual ranks on one GPU: the parity test)."""

    def __init__(self, rank: int = 0, size: int = 1, mode: str = "peer", group=None, device=None):
        assert mode in ("peer", "rccl", "loopback"), mode
        self.rank, self.size, self.mode, self.group, self.device = int(rank), int(size), mode, group, device
        self._opened = []
        self._comm: Optional[UmoeEpComm] = None

    @classmethod
    def from_dist(cls, mode: str = "peer", group=None, device=None) -> "EpLink":
        return cls(dist.get_rank(group), dist.get_world_size(group), mode, group, device)

    def connect(self, engine_handle):
        """Collective over `group` (every rank calls it with its own freshly created engine)."""
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        if self.mode == "loopback":
            L.check(lib.umoe_engine_ep_connect(engine_handle, None, None, L.EP_LOOPBACK), "umoe_engine_ep_connect")
            return
        if self.mode == "rccl":
            if self._comm is None:
                self._comm = UmoeEpComm(self.group, self.device)
            L.check(lib.umoe_engine_ep_connect(engine_handle, None, self._comm.h, L.EP_RCCL), "umoe_engine_ep_connect")
            return
        base, nbytes = C.c_void_p(), C.c_size_t()
        try:
            L.check(lib.umoe_engine_ep_region(engine_handle, C.byref(base), C.byref(nbytes)), "umoe_engine_ep_region")
            raw = (C.c_char * 64)()
            L.check(lib.umoe_ep_ipc_export(base, C.cast(raw, C.c_void_p)), "umoe_ep_ipc_export")
            mine = (bytes(raw), int(nbytes.value), "")
        except Exception as e:
            mine = (b"", 0, f"rank {self.rank}: {e!r}")
        every = [None] * self.size
        dist.all_gather_object(every, mine, group=self.group)
        bad = [m[2] for m in every if m[2]]
        if bad:
            raise L.UmoeError("expert-parallel connect failed: " + "; ".join(bad))
        peers = (C.c_void_p * L.MAX_EP)()
        problem = ""
        try:
            for p, (h, nb, _) in enumerate(every):
                if nb != mine[1]:
                    raise L.UmoeError(f"expert-parallel engines disagree on the exchange region size (rank {p}: {nb} != {mine[1]})")
                if p == self.rank:
                    peers[p] = base.value
                    continue
                ptr = C.c_void_p()
                hb = (C.c_char * 64).from_buffer_copy(h)
                L.check(lib.umoe_ep_ipc_open(C.cast(hb, C.c_void_p), C.byref(ptr)), f"umoe_ep_ipc_open (rank {p})")
                self._opened.append(ptr)
                peers[p] = ptr.value
            L.check(lib.umoe_engine_ep_connect(engine_handle, peers, None, L.EP_PEER), "umoe_engine_ep_connect")
        except Exception as e:           # agree on the outcome together: a rank that raised alone would leave its peers waiting
            problem = f"rank {self.rank}: {e!r}"
        outcomes = [None] * self.size
        dist.all_gather_object(outcomes, problem, group=self.group)   # (also the barrier: nobody steps before every rank has mapped every region)
        bad = [o for o in outcomes if o]
        if bad:
            raise L.UmoeError("expert-parallel connect failed: " + "; ".join(bad))

    @staticmethod
    def local_mesh(engine_handles):
        """Virtual ranks: engines of one process on one GPU store into each other's regions directly."""
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        bases = []
        for h in engine_handles:
            base, nbytes = C.c_void_p(), C.c_size_t()
            L.check(lib.umoe_engine_ep_region(h, C.byref(base), C.byref(nbytes)), "umoe_engine_ep_region")
            bases.append(base.value)
        for h in engine_handles:
            peers = (C.c_void_p * L.MAX_EP)(*bases)
            L.check(lib.umoe_engine_ep_connect(h, peers, None, L.EP_PEER), "umoe_engine_ep_connect")

    def close(self):
        from . import _lib as L
        for ptr in self._opened:
            L.lib().umoe_ep_ipc_close(ptr)
        self._opened = []
        if self._comm is not None:
            self._comm.close()
            self._comm = None

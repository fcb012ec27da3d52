"""Python mirror of the reference's accelerated batch interface for the hot path.

`HipBatchProcessor` keeps the four-method shape of the reference's CUDABatchProcessor
(src/cuda/cudabatch.hpp:39-59: addWindow / hasWindows / generateConsensus / reset) and `Window`
mirrors racon::Window (src/window.hpp:27-55: createWindow / add_layer / consensus).  All compute
goes through libvechat_hip.so's C ABI; nothing here computes consensus on the CPU.
"""
import ctypes as C

import numpy as np

from . import capi


class VcError(RuntimeError):
    pass


MAX_NODES = 59968      # 16-bit node ids, capacities are multiples of 64 (vc_submit)
MAX_EDGES = 32000


class HipContext:
    """Owns a vc_ctx: one device, two batch slots -- `submit(b1)` while b0 runs, `collect()` hands out the oldest run
    (include/vechat_hip.h, "Pipelining inside one context")."""

    def __init__(self, params=None, pipeline=None, reserve=None, **kw):
        """pipeline: None = the library's default (lock-step, or what VC_PIPE says); True / False = the persistent build
        pipeline on / off (vc_set_pipeline); a (forward_waves, backtrack_waves) pair also sizes its two kernels.
        reserve: None = workspaces are allocated under the first batch; a byte count (0 = the default budget) = allocated now,
        in one piece, and laid out per batch without further allocations (vc_reserve)."""
        self.lib = capi.load_hip()
        self.params = params or capi.default_params(**kw)
        h = C.c_void_p()
        rc = self.lib.vc_create(C.byref(h), C.byref(self.params))
        if rc != 0:
            raise VcError(f"vc_create failed ({rc}): {self.lib.vc_last_error(None).decode()}")
        self.h = h
        self._batch = None
        if pipeline is not None:
            fw, bw = pipeline if isinstance(pipeline, tuple) else (0, 0)
            self._chk(self.lib.vc_set_pipeline(self.h, 1 if pipeline else 0, fw, bw), "vc_set_pipeline")
        if reserve is not None:
            self._chk(self.lib.vc_reserve(self.h, int(reserve)), "vc_reserve")

    @classmethod
    def in_background(cls, **kw):
        """-> a future of a context: device start-up and vc_reserve run on a thread while the caller parses its input (the calls
        release the GIL); `.result()` when the first batch is ready."""
        from concurrent.futures import ThreadPoolExecutor
        ex = ThreadPoolExecutor(1)
        fut = ex.submit(lambda: cls(**kw))
        ex.shutdown(wait=False)
        return fut

    def reserve(self, nbytes=0):
        """vc_reserve: the workspaces' memory in one piece, now (0 = the default budget)."""
        self._chk(self.lib.vc_reserve(self.h, int(nbytes)), "vc_reserve")

    def set_polish_params(self, **kw):
        """vc_set_polish_params: overload (mode), thresholds, prune rounds, trim, window type and scores of this live context -- what
        differs between the driver's rounds; device, capacities and streams stay as created, the workspaces stay warm."""
        p = capi.VcParams.from_buffer_copy(self.params)
        for k, v in kw.items():
            if k not in ("mode", "min_confidence", "min_support", "num_prune", "trim", "window_type", "match", "mismatch", "gap",
                         "sw_match", "sw_mismatch", "sw_gap"):
                raise ValueError(f"{k} is fixed at vc_create")
            setattr(p, k, v)
        self._chk(self.lib.vc_set_polish_params(self.h, C.byref(p)), "vc_set_polish_params")
        self.params = p

    def release(self):
        """vc_release: workspaces (and a reservation) back to the device.  A staged batch that has not run can no longer be run -- submit it again."""
        self._chk(self.lib.vc_release(self.h), "vc_release")

    def set_window_type(self, window_type):
        self.params.window_type = int(window_type)
        self._chk(self.lib.vc_set_window_type(self.h, int(window_type)), "vc_set_window_type")

    def close(self):
        if getattr(self, "h", None):
            self.lib.vc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise VcError(f"{what} failed ({rc}): {self.lib.vc_last_error(self.h).decode()}")

    def submit(self, batch: capi.Batch):
        self._batch = batch                     # keep the numpy arrays alive during the H2D copies
        vb = batch.as_struct()
        self._chk(self.lib.vc_submit(self.h, C.byref(vb)), "vc_submit")

    def run(self):
        self._chk(self.lib.vc_run(self.h), "vc_run")

    def sync(self):
        self._chk(self.lib.vc_sync(self.h), "vc_sync")

    def collect(self):
        """-> (list of consensus bytes per window, status array) of the oldest run not yet collected"""
        nw = C.c_uint32(0)
        self._chk(self.lib.vc_result_windows(self.h, C.byref(nw)), "vc_result_windows")
        n = int(nw.value)
        size = C.c_uint64(0)
        self._chk(self.lib.vc_result_size(self.h, C.byref(size)), "vc_result_size")
        cons = np.zeros(max(int(size.value), 1), np.uint8)
        off = np.zeros(n + 1, np.uint64)
        status = np.zeros(n, np.uint8)
        r = capi.VcResult(off.ctypes.data_as(C.POINTER(C.c_uint64)), cons.ctypes.data_as(C.POINTER(C.c_uint8)),
                          cons.size, status.ctypes.data_as(C.POINTER(C.c_uint8)))
        self._chk(self.lib.vc_collect(self.h, C.byref(r)), "vc_collect")
        out = [cons[int(off[w]):int(off[w + 1])].tobytes() for w in range(n)]
        return out, status

    def stage_digest(self, kind, index, window=0):
        """Test hook: run the submitted (single-chunk) batch up to a stage and digest one window's graph / last alignment
        there -- [nodes, edges, hash(nodes), hash(edges), pairs, hash(pairs)] in the oracle's record format."""
        self._chk(self.lib.vc_debug_stop_after(self.h, 0 if kind == 4 else kind, index), "vc_debug_stop_after")
        try:
            self.run(); self.sync()
            out = (C.c_uint64 * 8)()
            self._chk(self.lib.vc_debug_stage_digest(self.h, window, 1 if kind in (1, 4) else 0, out), "vc_debug_stage_digest")
        finally:
            self.lib.vc_debug_stop_after(self.h, 0, 0)
        return [int(x) for x in out[2:8]]

    def errinfo(self):
        n = self._batch.n_windows
        out = np.zeros(n, np.uint32)
        self._chk(self.lib.vc_debug_errinfo(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "vc_debug_errinfo")
        return [(int(x) >> 16, int(x) & 0xFFFF) for x in out]

    def stats(self):
        s = capi.VcStats()
        self._chk(self.lib.vc_get_stats(self.h, C.byref(s)), "vc_get_stats")
        d = dict(cells=int(s.cells), alignments=int(s.alignments), dp_rows=int(s.dp_rows), far_row_reads=int(s.far_row_reads), trace_steps=int(s.trace_steps), trace_spec=int(s.trace_spec), trace_rounds=int(s.trace_rounds),
                 max_nodes=int(s.max_nodes), max_edges=int(s.max_edges), chunk_windows=int(s.chunk_windows), n_streams=int(s.n_streams),
                 band_redo=int(s.band_redo), device_bytes=int(s.device_bytes),
                 fwd_sclk_mhz=(100.0 * int(s.fwd_shader_cycles) / int(s.fwd_wall_ticks)) if int(s.fwd_wall_ticks) else None,
                 kernels={})
        for i in range(s.n_classes):
            d["kernels"][s.names[i].value.decode()] = dict(ms=float(s.ms[i]), launches=int(s.launches[i]), busy_ms=float(s.busy_ms[i]))
        return d

    def consensus_batched(self, batch: capi.Batch, batch_windows=32768, first=8192, retry_overflow=True, fill=None):
        """The loop of include/vechat_hip.h over one large batch: slices are queued behind each other in this context -- the copy-in of slice
        i + 1 and the copy-out of slice i - 1 run while slice i computes (the reference's accelerated polisher fills the next batch while one
        computes, src/cuda/cudapolisher.cpp:246-277).  A small first slice starts the device early.  Same bytes as consensus(); windows
        that outgrow the capacity estimate are retried the same way.
        fill(lo, hi): the batch is laid out but its windows are not written yet (WindowBuilder.build_streaming): the windows of a slice
        are written right before the slice is submitted, i.e. while the slices in front of it run."""
        n = batch.n_windows
        if n <= first + batch_windows // 2:
            if fill is not None:
                fill(0, n)
            return self.consensus(batch, retry_overflow=retry_overflow)
        sizes = [first]
        rem = (n - first) % batch_windows
        if rem:
            sizes.append(rem)                   # the odd remainder early (it runs beside full slices), full slices to the end
        sizes += [batch_windows] * ((n - sum(sizes)) // batch_windows)
        cuts = [0]
        for k in sizes:
            cuts.append(cuts[-1] + k)
        def part(i):
            if fill is not None:
                fill(cuts[i], cuts[i + 1])
            return batch.slice(cuts[i], cuts[i + 1])              # (offsets are rebased here: after the fill)
        outs = []
        self.submit(part(0)); self.run()
        for i in range(1, len(sizes)):
            self.submit(part(i)); self.run()
            outs.append(self.collect())
        outs.append(self.collect())
        cons = [x for o in outs for x in o[0]]
        status = np.concatenate([o[1] for o in outs])
        over = [w for w in range(n) if int(status[w]) == capi.VC_WIN_OVERFLOW]
        if retry_overflow and over:
            c2, s2 = self.consensus(batch.select(over), retry_overflow=True)
            for k, w in enumerate(over):
                cons[w], status[w] = c2[k], s2[k]
        return cons, status

    def consensus(self, batch: capi.Batch, retry_overflow=True):
        """submit + run + collect.  Windows whose graph outgrew the capacity estimate come back as
        VC_WIN_OVERFLOW with no bytes; with retry_overflow they are resubmitted (on the device, never on
        the CPU) in a context with doubled capacities, as INTEGRATION.md section 4 describes."""
        self.submit(batch)
        self.run()
        self.sync()
        cons, status = self.collect()
        over = [w for w in range(batch.n_windows) if int(status[w]) == capi.VC_WIN_OVERFLOW]
        if retry_overflow and over:
            st = self.stats()
            p = capi.VcParams.from_buffer_copy(self.params)
            p.max_nodes = min(2 * st["max_nodes"], MAX_NODES)
            p.max_edges = min(2 * st["max_edges"], MAX_EDGES)
            if p.max_nodes > st["max_nodes"] or p.max_edges > st["max_edges"]:
                self.lib.vc_release(self.h)             # this context's workspaces (up to 60 % of the device) go back first: the retry plans on what is free
                try:
                    sub = HipContext(params=p)
                except VcError:
                    return cons, status
                try:
                    c2, s2 = sub.consensus(batch.select(over), retry_overflow=True)
                except VcError:
                    return cons, status              # the larger capacities do not fit this device: the windows stay VC_WIN_OVERFLOW
                finally:
                    sub.close()
                for k, w in enumerate(over):
                    cons[w], status[w] = c2[k], s2[k]
        return cons, status


class Window:
    """racon::Window (src/window.hpp:27-55): backbone + layers as borrowed byte strings."""

    def __init__(self, id_, rank, window_type, backbone, quality):
        if len(backbone) == 0 or len(backbone) != len(quality.rstrip(b"\0")[:len(backbone)]):
            raise ValueError("[racon::createWindow] error: empty backbone sequence/unequal quality length!")
        self.id, self.rank, self.type = id_, rank, window_type
        self.sequences = [backbone]
        self.qualities = [quality]          # may be longer than the backbone (pointer into a longer buffer)
        self.positions = [(0, 0)]
        self.consensus = b""

    def add_layer(self, sequence, quality, begin, end):
        """Window::add_layer (src/window.cpp:47-72)."""
        if len(sequence) == 0 or begin == end:
            return
        if quality is not None and len(sequence) != len(quality):
            raise ValueError("[racon::Window::add_layer] error: unequal quality size!")
        L = len(self.sequences[0])
        if begin >= end or begin > L or end > L:
            raise ValueError("[racon::Window::add_layer] error: layer begin and end positions are invalid!")
        self.sequences.append(sequence)
        self.qualities.append(quality)
        self.positions.append((begin, end))


def create_window(id_, rank, window_type, backbone, quality):
    """racon::createWindow (src/window.cpp:17-31)."""
    return Window(id_, rank, window_type, backbone, quality)


class HipBatchProcessor:
    """addWindow / hasWindows / generateConsensus / reset, as src/cuda/cudabatch.hpp:39-59."""

    def __init__(self, ctx: HipContext, max_windows=1 << 20):
        self.ctx = ctx
        self.max_windows = max_windows
        self.windows = []

    def addWindow(self, window: Window) -> bool:
        if len(self.windows) >= self.max_windows:
            return False
        self.windows.append(window)
        return True

    def hasWindows(self) -> bool:
        return bool(self.windows)

    def reset(self):
        self.windows = []

    def generateConsensus(self):
        """Runs the batch; sets window.consensus; returns the list of bools
        Window::generate_consensus would have returned."""
        host = self.ctx.lib
        wins, fasta = [], []
        for w in self.windows:
            L = len(w.sequences[0])
            q0 = w.qualities[0]
            fasta.append(host.vc_backbone_is_fasta(q0, L))
            quals = [q0[:L]] + w.qualities[1:]
            wins.append((w.sequences, quals, [p[0] for p in w.positions], [p[1] for p in w.positions]))
        batch = capi.Batch.from_windows(wins, fasta, host=host)
        cons, status = self.ctx.consensus(batch)
        flags = []
        for w, c, s in zip(self.windows, cons, status):
            if s > capi.VC_WIN_UNPOLISHED:
                raise VcError(f"window {w.id}/{w.rank}: device status {int(s)} (no CPU fallback)")
            w.consensus = c
            flags.append(s == capi.VC_WIN_OK)
        return flags

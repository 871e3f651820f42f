"""Row-tile sharding of one frame across the GPUs of a node.

The edit chain is pointwise and the demosaic is a 5x5 stencil, so a frame cut into contiguous row
bands needs exactly one exchange step: each rank sends its first/last two Bayer rows to the rank
above/below (2 x W x 2 B per direction) and receives theirs into the two halo rows that bracket its
band buffer.  Neighbour-only send/recv over RCCL (torch.distributed, backend "nccl" on ROCm) — no
all-reduce/all-gather exists on this path.  torch is plumbing here (device memory + the process
group); the kernels are launched through the C ABI on raw device pointers.

Band buffer layout (matches include/rpf.h rpf_dev_develop): (rows + 4) x W u16, row r holds frame
row y0 - 2 + r; rows 0..1 and rows+2..rows+3 are halo.  Halo rows outside the frame are never read.
"""
from __future__ import annotations

from typing import List, Tuple

HALO = 2


def split_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous bands [(y0, rows)] with even boundaries (so every band starts on the CFA phase of
    row 0) and sizes differing by at most 2 rows."""
    if world < 1 or height < 2 * world:
        raise ValueError("need at least two rows per band")
    cuts = [((height * r // world) // 2) * 2 for r in range(world)] + [height]
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def halo_ops(band, rank: int, world: int, group=None) -> list:
    """The point-to-point operations of one halo exchange of `band` (views of it, no copies).  The list can be
    kept and handed to `torch.distributed.batch_isend_irecv` every step while `band` stays the same tensor."""
    import torch
    import torch.distributed as dist

    rows = band.shape[0] - 2 * HALO
    if rows < HALO:
        raise ValueError("a band must have at least 2 rows to feed its neighbours' halos")
    # RCCL/NCCL has no 16-bit integer datatype: ship the rows as bytes
    raw = band.view(torch.uint8) if band.dtype != torch.uint8 else band
    ops = []
    if rank > 0:  # neighbour above: my first two rows are its bottom halo
        ops.append(dist.P2POp(dist.isend, raw[HALO:2 * HALO], rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, raw[0:HALO], rank - 1, group))
    if rank < world - 1:  # neighbour below
        ops.append(dist.P2POp(dist.isend, raw[rows:rows + HALO], rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, raw[rows + HALO:rows + 2 * HALO], rank + 1, group))
    return ops


def exchange_halos(band, rank: int, world: int, group=None, ops=None) -> None:
    """Fill `band`'s halo rows from the neighbouring ranks' edge rows.

    band: torch tensor (rows + 4, W), any 2-byte dtype, contiguous, on the rank's device (or CPU for
    gloo).  One batch of point-to-point ops; returns when the halos are usable on the current stream.
    `ops`: the list `halo_ops` built for this band earlier (saves rebuilding it every step).
    """
    if world == 1:
        return
    import torch.distributed as dist

    for req in dist.batch_isend_irecv(ops if ops is not None else halo_ops(band, rank, world, group)):
        req.wait()


def make_band_buffer(cfa_band, device=None):
    """(rows, W) uint16 numpy -> (rows + 4, W) int16 torch tensor with zeroed halos (bit-identical
    storage; torch has no arithmetic on uint16 but none is needed)."""
    import numpy as np
    import torch

    rows, w = cfa_band.shape
    pad = np.zeros((rows + 2 * HALO, w), np.uint16)
    pad[HALO:HALO + rows] = cfa_band
    t = torch.from_numpy(pad.view(np.int16))
    return t.to(device) if device is not None else t


class _PartEditor:
    """What BandDeveloper hands to the `configure` callback: the editor of ONE part (row range) of the band.

    Every attribute is the part's own PhotoEditor's, except add_mask: the callback describes the band, so a
    band-sized mask (band_rows x W) is cut to the part's rows here (lib.rs:481-499 takes a whole-image mask; the
    band is this rank's image).  A mask that already has the part's size is passed through unchanged.
    """

    def __init__(self, ed, off, n, band_rows, width):
        self._ed, self._off, self._n, self._band_rows, self._w = ed, off, n, band_rows, width

    def __getattr__(self, name):
        return getattr(self._ed, name)

    def add_mask(self, name, mask_data):
        import numpy as np

        m = np.asarray(mask_data, dtype=np.float32)
        if m.size == self._band_rows * self._w and self._band_rows != self._n:
            m = m.reshape(self._band_rows, self._w)[self._off:self._off + self._n]
        self._ed.add_mask(name, m)


class BandDeveloper:
    """One rank's band of a row-tiled frame through the fused kernel, with the halo exchange hidden
    behind the interior rows.

    Only the first/last EDGE rows of a band read halo rows; the interior never does.  Per step (schedule
    "side", the default): the interior tile is launched on the current stream; a side stream, gated on what the
    current stream held before that launch, exchanges the 2-row halos with the neighbours (RCCL send/recv) and
    runs the two EDGE-row tiles while the interior is still computing; the current stream then waits for the
    side stream.  Measured on one MI355X without neighbours (tools/host_overhead.py): +1 % over a single
    launch; "serial" (exchange started first, everything else on one stream) +6 %, because the two tiny edge
    kernels then sit between consecutive interiors; "pipelined" (no per-step join: call join() before reading
    the output) +0 %.  With world == 1 (or overlap=False) it is one launch.
    All three tiles are views of the same band / output buffers (rpf_dev_develop takes raw pointers).

    `configure(ed)` sets up the edit state; it is called once per part with a view of that part's editor on which
    `add_mask(name, mask)` accepts a mask of the whole BAND (rows x W) and keeps the part's rows of it, so the same
    callback serves the single launch and the three-launch split.
    """

    EDGE = 8  # rows per edge tile = one work item of the kernel (64 columns x 8 rows)

    def __init__(self, gpu, frame_width, frame_height, y0, rows, raw, configure, rank=0, world=1, overlap=True,
                 force_split=False, schedule="side", packed=None):
        import torch

        from .editor import PhotoEditor

        self.raw = raw.to_c() if hasattr(raw, "to_c") else raw  # converted once, not per launch
        # packed = (bits, row_stride_bytes, order): the band buffer holds bit-packed rows (uint8, (rows + 4) x row_stride)
        # and the fused kernel unpacks them itself (rpf_dev_develop_packed)
        self.packed = packed
        self.rank, self.world = rank, world
        self._ops_for = None  # (band tensor id, data_ptr) the cached P2P op list belongs to
        self._ops = None
        self.w, self.rows = frame_width, rows
        # force_split: the three-launch schedule without neighbours (measuring its host-side cost on one GPU)
        self.split = ((overlap and world > 1) or force_split) and rows >= 4 * self.EDGE
        self._torch = torch
        e = self.EDGE
        parts = [(0, rows)] if not self.split else [(e, rows - 2 * e), (0, e), (rows - e, e)]
        self.parts = []
        for off, n in parts:  # (row offset inside the band, rows): interior first
            ed = PhotoEditor.for_tile(gpu, frame_width, frame_height, y0 + off, n)
            configure(_PartEditor(ed, off, n, rows, frame_width))
            self.parts.append((off, n, ed))
        self.schedule = schedule
        self.side = torch.cuda.Stream() if self.split and schedule != "serial" else None
        self._gate = torch.cuda.Event() if self.side is not None else None

    def join(self):
        """Make the current stream wait for edge tiles still running on the side stream ("pipelined" schedule)."""
        if self.side is not None:
            self._torch.cuda.current_stream().wait_stream(self.side)

    def step(self, band, out, out_format=0, bytes_per_px=16, exchange=True, events=None):
        """band: (rows + 4, W) 2-byte tensor with halo rows (packed: (rows + 4, row_stride) uint8); out: rows x W pixels.  `events`: optional
        (start, end) CUDA events recorded around the dominant (interior / whole-band) launch."""
        import torch.distributed as dist

        cur = self._torch.cuda.current_stream()
        stream = cur.cuda_stream
        bptr, optr = band.data_ptr(), out.data_ptr()
        row_in, row_out = (self.packed[1] if self.packed else self.w * 2), self.w * bytes_per_px

        def launch_on(part, on_stream):
            off, _, ed = part
            if self.packed:
                bits, stride, order = self.packed
                ed.dev_develop_packed(bptr + off * row_in, stride, bits, self.raw, optr + off * row_out, out_format,
                                      on_stream, order)
            else:
                ed.dev_develop(bptr + off * row_in, self.raw, optr + off * row_out, out_format, on_stream)

        def launch(part):
            launch_on(part, stream)

        if self.side is not None:
            # exchange + edge tiles on the side stream, gated on what the current stream held BEFORE the interior
            torch = self._torch
            self._gate.record(cur)
            if events:
                events[0].record(cur)
            launch(self.parts[0])
            if events:
                events[1].record(cur)
            self.side.wait_event(self._gate)
            with torch.cuda.stream(self.side):
                sstream = self.side.cuda_stream
                if exchange and self.world > 1:
                    key = (id(band), bptr)
                    if self._ops_for != key:
                        self._ops, self._ops_for = halo_ops(band, self.rank, self.world), key
                    for r in dist.batch_isend_irecv(self._ops):
                        r.wait()
                for part in self.parts[1:]:
                    launch_on(part, sstream)
            if self.schedule != "pipelined":
                cur.wait_stream(self.side)
            return
        reqs = ()
        if exchange and self.world > 1:
            key = (id(band), bptr)
            if self._ops_for != key:
                self._ops, self._ops_for = halo_ops(band, self.rank, self.world), key
            reqs = dist.batch_isend_irecv(self._ops)
            if not self.split:
                for r in reqs:
                    r.wait()
                reqs = ()
        if events:
            events[0].record(cur)
        launch(self.parts[0])
        if events:
            events[1].record(cur)
        if not self.split:
            return
        for r in reqs:  # the current stream waits for the halo rows; the interior is already queued ahead of that
            r.wait()
        launch(self.parts[1])
        launch(self.parts[2])


class ShardedDeveloper:
    """One rank's band through the library's own sharded step (include/rpf.h rpf_shard_develop): ONE C call per frame
    enqueues the RCCL halo exchange and the band's launch on the step's stream; step_async hands the same step to the shard's
    own thread (rpf_shard_develop_async: two ShardedDevelopers on two shards and two streams, frames alternating, are the
    two-pipeline schedule).  The measured N > 1 configuration of bench.py; BandDeveloper (torch.distributed) remains as the
    rehearsal path for gloo and as the fallback when RCCL cannot be initialised inside the library.

    `shard`: a rawphotoforge_amd.Shard (None: a one-rank shard without neighbours is created).
    `configure(ed)`: sets up the edit state of the band's editor; `add_mask` takes band-sized masks (rows x W).
    """

    def __init__(self, gpu, frame_width, frame_height, y0, rows, raw, configure, shard=None, packed=None):
        from . import _native as N
        from .editor import PhotoEditor, Shard

        self.raw = raw.to_c() if hasattr(raw, "to_c") else raw
        self.packed = packed  # (bits, row_stride_bytes, order)
        self.shard = shard if shard is not None else Shard(gpu, None, 0, 1)
        self.w, self.rows = frame_width, rows
        self.ed = PhotoEditor.for_tile(gpu, frame_width, frame_height, y0, rows)
        configure(self.ed)
        self._no_exchange = N.SHARD_NO_EXCHANGE
        self.split = False
        # bench.py's accounting: (row offset, rows, editor) of the launch its kernel events bracket
        self.parts = [(0, rows, self.ed)]

    def join(self):
        """Nothing to join: the step is stream-ordered (BandDeveloper API compatibility)."""

    def step_async(self, band, out, out_format=0, exchange=True, stream=None, timed=False):
        """The step handed to the shard's own thread (rpf_shard_develop_async); `stream`: a raw stream handle (default: the
        current torch stream).  Call shard.drain(stream) (or flush()) before relying on the stream's contents."""
        from . import _native as N
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        self.shard.develop_async(self.ed, band.data_ptr(), self.raw, out.data_ptr(), out_format, stream,
                                 (0 if exchange else self._no_exchange) | (N.SHARD_TIMED if timed else 0), self.packed)

    def step(self, band, out, out_format=0, bytes_per_px=16, exchange=True, events=None, stream=None):
        """band: (rows + 4, W) 2-byte tensor with halo rows (packed: (rows + 4, row_stride) uint8); out: rows x W pixels.
        `events`: optional (start, end) CUDA events recorded on the launch stream around the WHOLE step (exchange + launch)."""
        if stream is None:
            import torch
            cur = torch.cuda.current_stream()
            stream = cur.cuda_stream
        else:
            cur = None
        if events and cur is not None:
            events[0].record(cur)
        self.shard.develop(self.ed, band.data_ptr(), self.raw, out.data_ptr(), out_format, stream,
                           0 if exchange else self._no_exchange, self.packed)
        if events and cur is not None:
            events[1].record(cur)

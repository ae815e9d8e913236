"""Row-partitioned multi-GPU rendering: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

The reference has no distributed layer at all; this is the north-star design:
rank g renders image rows ``[row0, row1)`` of every camera, the RGBA bands are reassembled with ONE
all-gather, and the backward needs exactly two small exchanges:

* visibility flags (P bytes, MAX): ``rs[n]`` is the median radius of the *globally* visible points
  (rasterizer.py:885-888), so every rank must see the union before ``dss_backward_radius``;
* per-point gradient partials ``(P,3)`` and ``(P,C)`` (SUM): every rank accumulates the pixels of its
  band only; the sum over bands is the full gradient.

Point parameters are replicated (they are the model); nothing else crosses the links.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


class RowPartition:
    """Image rows owned by ``rank``.  Three layouts:

    * equal bands (default): ``S`` rounded up to a multiple of ``world_size``; the last band may be shorter; one
      fixed-size all-gather reassembles them;
    * ``bounds`` (``world_size + 1`` non-decreasing row indices from 0 to S, see ``balanced_bounds``): contiguous bands
      that follow a load estimate; ``band`` is then the largest band (the padded exchange size);
    * ``cyclic=True``: TILE-ROW-CYCLIC -- rank g owns the 8-row tile rows g, g + G, g + 2G, ... (``world_size`` a power
      of two).  Every rank gets the same mix of dense and empty screen regions, so the per-rank load is balanced for
      any scene without a load estimate, the bands have equal size when ``S`` is a multiple of ``8 * world_size`` (one
      fixed-size all-gather, no padding; otherwise they differ by at most one tile row and travel padded to ``band``),
      and nothing has to be re-balanced when the object moves.  ``rows`` is then
      the triple ``(8 * rank, S, world_size)`` the fused entry points take (``row_cycle``, include/dss_hip.h); band
      tensors hold the owned rows in image order."""

    def __init__(self, image_size: int, world_size: int = 1, rank: int = 0, bounds=None, cyclic: bool = False):
        if not (0 <= rank < world_size):
            raise ValueError("rank %d outside world of %d" % (rank, world_size))
        self.S, self.world_size, self.rank = int(image_size), int(world_size), int(rank)
        self.cyclic = bool(cyclic) and self.world_size > 1
        if self.cyclic:
            G = self.world_size
            if bounds is not None:
                raise ValueError("cyclic and bounds are exclusive")
            if G & (G - 1):
                raise ValueError("a tile-row-cyclic partition needs a power-of-two world size, got %d" % G)
            self._bounds = None
            self.band = max(max(len(self.row_indices(g)) for g in range(G)), 1)
            self.uniform = False          # (not a contiguous band: the contiguous helpers refuse it)
            self.row0, self.row1 = min(8 * self.rank, self.S), self.S
            return
        if bounds is None:
            self.band = -(-self.S // self.world_size)  # ceil
            self._bounds = [min(r * self.band, self.S) for r in range(self.world_size + 1)]
            self.uniform = True
        else:
            b = [int(x) for x in bounds]
            if len(b) != self.world_size + 1 or b[0] != 0 or b[-1] != self.S or any(b[i] > b[i + 1] for i in range(len(b) - 1)):
                raise ValueError("bounds must be %d non-decreasing row indices from 0 to %d, got %r" % (self.world_size + 1, self.S, b))
            self._bounds = b
            self.band = max(max(b[i + 1] - b[i] for i in range(self.world_size)), 1)
            self.uniform = False
        self.row0, self.row1 = self._bounds[self.rank], self._bounds[self.rank + 1]

    @property
    def rows(self):
        """what the kernels take as `rows`: (row0, row1) for a contiguous band, (row0, row1, cycle) for a cyclic one"""
        return (self.row0, self.row1, self.world_size) if self.cyclic else (self.row0, self.row1)

    @property
    def n_rows(self) -> int:
        return len(self.row_indices()) if self.cyclic else self.row1 - self.row0

    def bounds(self, rank: int) -> Tuple[int, int]:
        if self.cyclic:
            raise ValueError("a cyclic partition has no contiguous bounds")
        return self._bounds[rank], self._bounds[rank + 1]

    def row_indices(self, rank: Optional[int] = None):
        """image rows owned by `rank` (default: this rank), in band order"""
        g = self.rank if rank is None else rank
        if self.cyclic:
            G = self.world_size
            return [t * 8 + i for t in range(g, -(-self.S // 8), G) for i in range(8) if t * 8 + i < self.S]
        r0, r1 = self._bounds[g], self._bounds[g + 1]
        return list(range(r0, r1))

    def gather_index(self):
        """for every image row r: its position in the all-gathered (world_size * band, ...) buffer"""
        pos = [0] * self.S
        for g in range(self.world_size):
            for l, r in enumerate(self.row_indices(g)):
                pos[r] = g * self.band + l
        return pos

    def describe(self) -> str:
        if self.cyclic:
            return "tile-row-cyclic: rank g owns the 8-row tile rows g, g+%d, ... (%d rows each)" % (self.world_size, self.band)
        return ("equal bands of %d rows" % self.band) if self.uniform else ("bands %r" % (self._bounds,))

    def slice(self, full: torch.Tensor) -> torch.Tensor:
        """Own rows of a full-image tensor (N, S, ...) (a view for contiguous bands, a gather for the cyclic layout)."""
        if self.cyclic:
            if self.S % (8 * self.world_size) == 0:
                N = full.shape[0]
                t = full.reshape((N, self.S // (8 * self.world_size), self.world_size, 8) + tuple(full.shape[2:]))
                return t[:, :, self.rank].reshape((N, self.band) + tuple(full.shape[2:]))
            return full.index_select(1, torch.tensor(self.row_indices(), dtype=torch.int64, device=full.device))
        return full[:, self.row0:self.row1]


def balanced_bounds(row_weight, world_size: int, align: int = 8, min_rows: int = 8):
    """Band boundaries with (about) equal total ``row_weight`` per rank, multiples of ``align`` rows, every band at
    least ``min_rows`` high.  ``row_weight`` (S,) is any per-row load estimate that is identical on every rank
    (e.g. occupied pixels per row of a first render): equal ROW counts leave the middle ranks of a centred
    object with twice the work of the outer ones."""
    w = torch.as_tensor(row_weight, dtype=torch.float64).flatten().cpu()
    S, G = int(w.numel()), int(world_size)
    if G * min_rows > S:
        raise ValueError("cannot give %d ranks %d rows each out of %d" % (G, min_rows, S))
    cum = torch.cumsum(w + 1e-9, 0)
    total = float(cum[-1])
    bounds = [0]
    for g in range(1, G):
        target = total * g / G
        r = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64)).item()) + 1
        r = int(round(r / align)) * align
        r = max(r, bounds[-1] + min_rows)             # this band is at least min_rows high ...
        r = min(r, S - (G - g) * min_rows)            # ... and so can every later one be
        bounds.append(r)
    bounds.append(S)
    return bounds


def fitted_bounds(row_weight, samples, world_size: int, align: int = 8, min_rows: int = 8):
    """Band boundaries from MEASURED per-rank step times: ``samples`` = [(bounds, times), ...] of at least two different
    contiguous partitions (e.g. equal rows, then `balanced_bounds` of the occupancy).  Fits  t_rank = F + a * (row_weight
    summed over the rank's rows) + b * (rows of the rank)  with F, a, b >= 0 -- F is the work every rank repeats (setup of
    every splat, the medians), a the cost of covered pixels, b what a row costs covered or not -- and returns
    `balanced_bounds` of the per-row cost a * row_weight + b, together with (F, a, b).  Occupancy alone (b = 0) moved too
    many empty rows to the outer ranks at configs[3]/[4]: their rows are not free (tile lists, alpha plane, image bytes).
    EVERY RANK MUST PASS THE SAME ``samples`` (all-gather the measured times first): ranks that derive different bounds
    disagree about the band sizes of the exchange and hang in it.  `agree_on_bounds` broadcasts rank 0's result."""
    import itertools
    import numpy as np
    w = np.asarray(torch.as_tensor(row_weight, dtype=torch.float64).flatten().cpu())
    cum = np.concatenate([[0.0], np.cumsum(w)])
    A, y = [], []
    for bounds, times in samples:
        for r in range(len(bounds) - 1):
            A.append([1.0, cum[bounds[r + 1]] - cum[bounds[r]], float(bounds[r + 1] - bounds[r])])
            y.append(float(times[r]))
    A, y = np.asarray(A), np.asarray(y)
    if not (np.isfinite(y).all() and (y >= 0).all()):
        raise ValueError("fitted_bounds: the measured times must be finite and non-negative, got %r" % (y.tolist(),))
    scale = np.maximum(np.abs(A).max(axis=0), 1e-30)
    best = None
    for k in (3, 2, 1):                       # non-negative least squares over the subsets of {F, a, b}
        for cols in itertools.combinations(range(3), k):
            x = np.zeros(3)
            sol = np.linalg.lstsq(A[:, cols] / scale[list(cols)], y, rcond=None)[0] / scale[list(cols)]
            if (sol < 0).any():
                continue
            x[list(cols)] = sol
            err = float(((A @ x - y) ** 2).sum())
            if best is None or err < best[0] - 1e-12:
                best = (err, x)
    if best is None:
        raise ValueError("fitted_bounds: no non-negative fit of t = F + a * weight + b * rows exists for these samples "
                         "(are the times positive and finite?)")
    F, a, b = (float(v) for v in best[1])
    if a <= 0 and b <= 0:
        return balanced_bounds(np.ones_like(w), world_size, align, min_rows), (F, a, b)
    return balanced_bounds(a * w + b, world_size, align, min_rows), (F, a, b)


def rebalanced_bounds(bounds, times, fixed: float, align: int = 8, min_rows: int = 8):
    """One step of measured-time rebalancing of contiguous bands: every rank's time above the repeated work ``fixed`` (the F
    of `fitted_bounds`) is spread evenly over its rows -- a piecewise-constant cost per row, measured, whatever it is made
    of (tile lists, depth complexity, window sizes) -- and the boundaries are moved so that every rank gets the same share.
    ``times`` must be the same list on every rank (see `fitted_bounds`; `agree_on_bounds`)."""
    S, G = int(bounds[-1]), len(bounds) - 1
    w = torch.zeros(S, dtype=torch.float64)
    for r in range(G):
        rows = bounds[r + 1] - bounds[r]
        w[bounds[r]:bounds[r + 1]] = max(float(times[r]) - float(fixed), 1e-6) / max(rows, 1)
    return balanced_bounds(w, G, align, min_rows)


def agree_on_bounds(bounds, group=None, device=None):
    """Rank 0's band boundaries on every rank (a broadcast of world_size + 1 integers): the guard for bounds derived from
    locally measured quantities.  Without an initialised process group the input is returned as it is."""
    b = [int(x) for x in bounds]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return b
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.tensor(b, dtype=torch.int64, device=(device if device is not None else "cuda") if on_gpu else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return [int(x) for x in t.tolist()]


def gather_rows(band: torch.Tensor, part: RowPartition, group=None) -> torch.Tensor:
    """All-gather row bands ``(N, rows, S, ch)`` into the full image ``(N, S, S, ch)`` on every rank."""
    if part.world_size == 1:
        return band
    if not part.uniform:
        raise ValueError("gather_rows needs equal bands; use OverlappedExchange for load-balanced bounds")
    n, rows = band.shape[0], band.shape[1]
    if rows < part.band:  # short (or empty) last band: pad to the common size
        pad = band.new_zeros((n, part.band - rows) + tuple(band.shape[2:]))
        band = torch.cat([band, pad], dim=1)
    band = band.contiguous()
    out = band.new_empty((part.world_size,) + tuple(band.shape))
    if dist.get_backend(group) == "gloo":  # CPU tests; gloo has no all_gather_into_tensor
        dist.all_gather(list(out.unbind(0)), band, group=group)
    else:
        dist.all_gather_into_tensor(out, band, group=group)
    # (G, N, band, S, ch) -> (N, G*band, S, ch) -> crop
    full = out.permute(1, 0, 2, *range(3, out.dim())).reshape((n, part.world_size * part.band) + tuple(band.shape[2:]))
    return full[:, :part.S]


def gather_rows_and_visibility(band: torch.Tensor, visible: torch.Tensor, part: RowPartition, group=None):
    """ONE all-gather for both end-of-forward exchanges: the RGBA row bands and the per-band visibility
    flags travel in the same byte buffer (collectives here are latency-, not bandwidth-bound: fewer calls
    matter more than fewer bytes).  Returns (full image (N,S,S,ch), union of the visibility flags uint8 (P,))."""
    if part.world_size == 1:
        return band, visible
    if not part.uniform:
        raise ValueError("gather_rows_and_visibility needs equal bands; use OverlappedExchange for load-balanced bounds")
    n, rows = band.shape[0], band.shape[1]
    if rows < part.band:
        pad = band.new_zeros((n, part.band - rows) + tuple(band.shape[2:]))
        band = torch.cat([band, pad], dim=1)
    vis8 = visible.view(torch.uint8) if visible.dtype == torch.bool else visible
    nb = band.numel() * band.element_size()
    send = torch.cat([band.contiguous().view(-1).view(torch.uint8), vis8.reshape(-1)])
    out = send.new_empty((part.world_size, send.numel()))
    if dist.get_backend(group) == "gloo":
        dist.all_gather(list(out.unbind(0)), send, group=group)
    else:
        dist.all_gather_into_tensor(out, send, group=group)
    bands = out[:, :nb].contiguous().view(band.dtype).view((part.world_size,) + tuple(band.shape))
    full = bands.permute(1, 0, 2, *range(3, bands.dim())).reshape((n, part.world_size * part.band) + tuple(band.shape[2:]))
    vis_all = out[:, nb:].max(dim=0).values
    return full[:, :part.S], vis_all


class ForwardExchange:
    """Zero-copy variant of ``gather_rows_and_visibility``: the send buffer is laid out once
    ([band image | visibility flags], bytes); the forward kernel writes both outputs straight into it
    (``ops.render_forward(out_image=..., out_visible=...)``), so the step issues no packing kernels."""

    def __init__(self, part: RowPartition, n_images: int, channels: int, num_points: int, device):
        if not part.uniform:
            raise ValueError("ForwardExchange needs equal bands; use OverlappedExchange for load-balanced bounds")
        self.part = part
        self.shape = (n_images, part.band, part.S, channels)
        self.nb = n_images * part.band * part.S * channels * 4
        self.P = num_points
        self.send = torch.zeros(self.nb + num_points, dtype=torch.uint8, device=device)
        self.recv = torch.empty((part.world_size, self.nb + num_points), dtype=torch.uint8, device=device)
        rows = part.row1 - part.row0
        full_band = self.send[:self.nb].view(torch.float32).view(self.shape)
        self.image = full_band[:, :rows] if rows == part.band else None  # short last band: pack by copy
        self.visible = self.send[self.nb:]

    def exchange(self, band: Optional[torch.Tensor] = None, group=None):
        """-> (full image (N,S,S,ch), union of the visibility flags uint8 (P,))."""
        part = self.part
        if band is not None and (self.image is None or band.data_ptr() != self.image.data_ptr()):
            self.send[:self.nb].view(torch.float32).view(self.shape)[:, :band.shape[1]].copy_(band)
        if dist.get_backend(group) == "gloo":
            dist.all_gather(list(self.recv.unbind(0)), self.send, group=group)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=group)
        n = self.shape[0]
        bands = self.recv[:, :self.nb].contiguous().view(torch.float32).view((part.world_size,) + self.shape)
        full = bands.permute(1, 0, 2, 3, 4).reshape((n, part.world_size * part.band) + self.shape[2:])
        return full[:, :part.S], self.recv[:, self.nb:].max(dim=0).values


class OverlappedExchange:
    """End-of-forward exchanges arranged for overlap with the backward (bench.py, multi-GPU):

    * visibility flags (P bytes per rank) -- ONE in-place all-reduce (MAX over 0/1 bytes) on ``group``; the backward
      needs the union before its first kernel (``rs`` is the median radius of the GLOBALLY visible points), so this one
      is on the critical path and latency-bound;
    * RGBA bands (N * band * S * ch * 4 bytes per rank, 4 MB at 8 cameras x 512^2) -- an ASYNCHRONOUS
      all-gather on ``image_group``, a second process group (= its own RCCL communicator, so it neither
      orders with nor blocks the gradient all-reduce); it completes while the backward runs and is
      awaited at the end of the step (``finish``).

    The send buffer is laid out (band row, camera, col, channel): the rows a rank owns are contiguous for
    ALL cameras, so the gathered buffer is the full image in (row, camera, col, channel) order and is handed
    out as an (N, S, S, ch) strided view -- no reassembly copy (the (G, N, band, ...) layout needed a
    32 MB permute at 8 cameras).  ``image`` is the (N, rows, S, ch) view the forward kernel writes through
    (``ops.render_forward(out_image=...)`` takes camera / row strides)."""

    def __init__(self, part: RowPartition, n_images: int, channels: int, num_points: int, device, group=None,
                 image_group=None, force: bool = False, fold: bool = False):
        # force: build the second communicator and issue every collective even in a world of ONE rank (a one-GPU box
        # then executes the RCCL code path line for line: bench.py BENCH_FORCE_DIST=1)
        self.part, self.group = part, group
        self.image_group = image_group
        # `overlap`: the image bands travel on their own communicator, asynchronously.  If the second communicator cannot
        # be created (or the asynchronous collective later raises) the exchange degrades to ONE communicator and a
        # blocking all-gather -- slower, same result -- and says so (`overlap` False, `degraded` holds the reason): the
        # first multi-GPU run of a deployment must produce a diagnosable number rather than a stack trace.
        self.overlap, self.degraded = True, None
        if image_group is None and not fold and (part.world_size > 1 or force) and dist.is_initialized():
            try:
                self.image_group = dist.new_group()  # collective: every rank constructs its exchange
            except Exception as e:  # noqa: BLE001  (RCCL refused a second communicator)
                self.image_group, self.overlap = group, False
                self.degraded = "new_group failed: %s: %s" % (type(e).__name__, str(e)[:200])
            # every rank must take the same path (a rank that degraded alone would issue a different collective
            # sequence): agree on the minimum over the ranks
            ok = torch.tensor([1 if self.overlap else 0], dtype=torch.int32,
                              device=device if dist.get_backend(group) != "gloo" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0 and self.overlap:
                self.image_group, self.overlap = group, False
                self.degraded = "another rank could not create the second communicator"
        G, band, S = part.world_size, part.band, part.S
        # fold: TWO collectives per step instead of three -- the visibility flags ride in the image all-gather as `vrows`
        # extra rows behind every rank's band (one blocking all-gather on `group`, the union = a MAX over the G gathered
        # copies, the image rows put in place by one gather kernel).  The price: the backward waits for the whole image
        # exchange instead of a P-byte all-reduce, i.e. the image bytes move on the critical path.  Which form is faster
        # depends on the link time of the image bands against the latency of one more collective: bench.py measures both on
        # the ranks it runs on and keeps the faster (`config.dist.exchange`).
        self.fold = bool(fold)
        row_floats = n_images * S * channels
        self.vrows = -(-int(num_points) // (4 * row_floats)) if self.fold else 0
        self.stride_rows = band + self.vrows     # rows of one rank's chunk of the gathered buffer
        self.send_img = torch.zeros((band + self.vrows, n_images, S, channels), dtype=torch.float32, device=device)
        self.recv_img = torch.empty((G * (band + self.vrows), n_images, S, channels), dtype=torch.float32, device=device)
        rows = part.n_rows
        self.image = self.send_img[:rows].permute(1, 0, 2, 3)  # (N, rows, S, ch), strided
        if self.fold:
            # the forward kernel writes its flags straight into the tail of the send buffer
            self.visible = self.send_img[band:].reshape(-1).view(torch.uint8)[:num_points]
            self.union = torch.zeros(num_points, dtype=torch.uint8, device=device)
            self._gathered_flags = self.recv_img.view(G, band + self.vrows, row_floats)[:, band:]   # (G, vrows, row_floats)
        else:
            self.visible = torch.zeros(num_points, dtype=torch.uint8, device=device)
            self.union = None
        self.num_points = int(num_points)
        self._work = None
        on_gpu = torch.device(device).type == "cuda"
        self._side_img = torch.cuda.Stream(device=device) if on_gpu else None   # issue stream of the image collective
        self._fwd_event = torch.cuda.Event() if on_gpu else None
        self._fwd_done = self._issued_on = None
        self.late_image = False
        # load-balanced (unequal) bands travel padded to the largest one; the rows are put in place by ONE gather
        # kernel on the side stream as soon as the collective completes, i.e. still during the backward
        self.full_img = self.row_index = None
        if not part.uniform or self.fold:
            # image row r sits at gather_index()[r] of the gathered buffer (unequal bands travel padded to the largest
            # one; a cyclic partition interleaves the ranks' tile rows; folded flags sit between the ranks' bands)
            pos = part.gather_index()
            if self.vrows:
                pos = [(q // band) * self.stride_rows + q % band for q in pos]
            self.row_index = torch.tensor(pos, dtype=torch.int64, device=device)
            self.full_img = torch.empty((S, n_images, S, channels), dtype=torch.float32, device=device)

    @staticmethod
    def _all_gather(out2d: torch.Tensor, send: torch.Tensor, group, async_op: bool):
        if dist.get_backend(group) == "gloo":  # CPU tests; gloo has no all_gather_into_tensor
            return dist.all_gather(list(out2d.unbind(0)), send, group=group, async_op=async_op)
        return dist.all_gather_into_tensor(out2d, send, group=group, async_op=async_op)

    def start(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Issue both exchanges; returns the union of the visibility flags, uint8 (P,) (written into `out` if given: a static
        buffer that captured graphs read).  The visibility union goes FIRST: it is the one the backward waits for, and the
        GPU executes it while the host is still issuing the image collective (an RCCL call costs the host 15-20 us; measured
        at world size 1, profiles/r4_c_forced_dist_issue_order.txt)."""
        if self.fold:
            return self._start_folded(out)
        vis = self.start_visibility(out)
        self.start_image()
        return vis

    def _start_folded(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`fold`: ONE blocking all-gather of [band rows | visibility flags] on `group`; the union of the flags is a MAX over
        the gathered copies; the image rows are put in place by one gather kernel (on the side stream when there is one: it
        overlaps with the backward, which only needs the union)."""
        G = self.part.world_size
        self._all_gather(self.recv_img.view(G, -1), self.send_img.view(-1), self.group, False)
        flags = self._gathered_flags.reshape(G, -1).view(torch.uint8)[:, :self.num_points]
        union = self.union if out is None else out
        torch.amax(flags, dim=0, out=union)
        self._work = None
        if self._side_img is not None:
            self._side_img.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side_img):
                torch.index_select(self.recv_img, 0, self.row_index, out=self.full_img)
            self._issued_on = self._side_img
        else:
            torch.index_select(self.recv_img, 0, self.row_index, out=self.full_img)
            self._issued_on = None
        return union

    def start_visibility(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The critical exchange alone.  `late_image`: also records the point of the stream up to which the image bands are
        complete, for a `start_image` that is called after more work has been enqueued."""
        self._fwd_done = None
        if self.late_image and self._fwd_event is not None:
            self._fwd_done = self._fwd_event
            self._fwd_done.record()
        vis = self._union_visibility()
        if out is not None and out.data_ptr() != vis.data_ptr():
            out.copy_(vis)
            return out
        return vis

    def start_image(self) -> None:
        """The all-gather of the RGBA bands, asynchronous on its own communicator.  `late_image` (the caller issues it AFTER
        launching the backward, to hide the host's issue time behind it): the collective goes through a side stream that
        only waits for the end of the forward, so that it still overlaps with the backward -- measured slower at world size
        1 (the extra stream switch and event cost the host more than the reordering saves), hence off by default."""
        G = self.part.world_size
        side = self._side_img if (self.late_image and self._fwd_done is not None) else None
        if side is not None:
            side.wait_event(self._fwd_done)
        self._issued_on = side
        with (torch.cuda.stream(side) if side is not None else _null_context()):
            if self.overlap:
                try:
                    self._work = self._all_gather(self.recv_img.view(G, -1), self.send_img.view(-1), self.image_group, True)
                except Exception as e:  # noqa: BLE001  (asynchronous collective refused: blocking exchange from now on)
                    self.overlap, self._work = False, None
                    self.degraded = "async all_gather failed: %s: %s" % (type(e).__name__, str(e)[:200])
            if not self.overlap:
                self._all_gather(self.recv_img.view(G, -1), self.send_img.view(-1), self.image_group, False)
                self._work = None
                if self.row_index is not None:
                    torch.index_select(self.recv_img, 0, self.row_index, out=self.full_img)
        if self._work is not None and self.row_index is not None and self._side_img is not None:
            # unequal / cyclic bands: the rows are put in place on the side stream as soon as the collective completes
            if side is None:
                self._side_img.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side_img):
                self._work.wait()   # (the side stream waits for the collective, not the host)
                torch.index_select(self.recv_img, 0, self.row_index, out=self.full_img)
            self._issued_on = self._side_img

    def _union_visibility(self) -> torch.Tensor:
        """Union of the per-rank visibility flags, IN PLACE in `self.visible` (the buffer the forward kernel writes): ONE
        all-reduce (MAX over 0/1 bytes).  Round 3 all-gathered the flags and reduced them with a `max(dim=0)` kernel plus a
        copy into the graphs' static buffer: 47 us per step on the critical path at world size 1 (RCCL 2.26, measured through
        BENCH_FORCE_DIST) against 14 us for the gradient all-reduce of 25x the bytes -- latency, not bandwidth."""
        dist.all_reduce(self.visible, op=dist.ReduceOp.MAX, group=self.group)
        return self.visible

    def finish(self) -> torch.Tensor:
        """Wait for the image bands; returns the full render (N, S, S, ch) (strided view, no copy for equal bands)."""
        on_side = self._issued_on is not None
        if on_side:
            torch.cuda.current_stream().wait_stream(self._issued_on)   # the issue point and everything enqueued behind it
            self._issued_on = None
        if self._work is not None:
            self._work.wait()
            if self.row_index is not None and not on_side:  # CPU (tests)
                torch.index_select(self.recv_img, 0, self.row_index, out=self.full_img)
            self._work = None
        if self.row_index is not None:
            return self.full_img.permute(1, 0, 2, 3)
        return self.recv_img[:self.part.S].permute(1, 0, 2, 3)


class AlphaPlaneExchange:
    """All-gather of ONE channel of the bands' image gradients -- the occupancy (alpha) gradient -- into the dense
    (N, S, S) plane the owner form of the backward reads (`ops.render_backward(grad_occ_full=...)`).

    A rank that evaluates the loss on its own band (`band_image_loss`) only holds the gradient of its rows; the owned
    search windows reach ``rs`` beyond them (and, for tile-row-cyclic bands, into every other rank's rows).  Only the alpha
    channel is needed there (`rasterize_points_backward.cu:141-178` reads ``grad_occ``), so ``N S^2 4`` bytes cross the links
    in total -- a quarter of the RGBA gradient, an eighth of it per rank and link at 8 ranks; the RGB gradient stays where it
    was computed.  On the critical path (between the loss and the backward): issued on ``group``, blocking on the stream.
    Send layout (band row, camera, col), like the image exchange: equal contiguous bands of one camera ARE the plane."""

    def __init__(self, part: RowPartition, n_images: int, device, group=None):
        self.part, self.group, self.N = part, group, int(n_images)
        G, band, S = part.world_size, part.band, part.S
        self.send = torch.zeros((band, self.N, S), dtype=torch.float32, device=device)
        self.recv = torch.empty((G * band, self.N, S), dtype=torch.float32, device=device)
        # rows already in image order and one camera: the receive buffer is the plane
        self.direct = part.uniform and self.N == 1
        self.row_pos = self.plane = None
        if not self.direct:
            self.row_pos = torch.tensor(part.gather_index(), dtype=torch.int32, device=device)
            self.plane = torch.empty((self.N, S, S), dtype=torch.float32, device=device)

    def pack(self, g_band: torch.Tensor) -> None:
        """``g_band`` (N, rows, S, C+1), this rank's band of the image gradient: its last channel into the send buffer
        (one strided copy, (row, camera, col) order)"""
        self.send[:g_band.shape[1]].copy_(g_band[..., -1].permute(1, 0, 2))

    def gather(self) -> None:
        """the collective (blocking on the stream, on ``group``)"""
        G = self.part.world_size
        if dist.is_available() and dist.is_initialized():
            OverlappedExchange._all_gather(self.recv.view(G, -1), self.send.view(-1), self.group, False)
        else:
            self.recv.copy_(self.send)

    def plane_in_image_order(self) -> torch.Tensor:
        """-> (N, S, S) alpha gradient of all rows, rows in image order"""
        S = self.part.S
        if self.direct:
            return self.recv[:S].view(1, S, S)
        if self.recv.is_cuda:
            from . import ops
            return ops.gather_rows(self.recv, self.row_pos, self.N, S, S, out=self.plane)
        # (CPU tensors only occur in the gloo tests of the exchange logic)
        self.plane.copy_(torch.index_select(self.recv, 0, self.row_pos.long()).permute(1, 0, 2))
        return self.plane

    def exchange(self, g_band: torch.Tensor) -> torch.Tensor:
        """pack + gather + rows in image order"""
        self.pack(g_band)
        self.gather()
        return self.plane_in_image_order()


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class GatherRows(torch.autograd.Function):
    """Differentiable ``gather_rows``.  Every rank evaluates the same loss on the same full image, so
    the gradient of the local band is simply its slice of the full-image gradient (no collective)."""

    @staticmethod
    def forward(ctx, band, part, group):
        ctx.part = part
        return gather_rows(band, part, group)

    @staticmethod
    def backward(ctx, grad_full):
        return ctx.part.slice(grad_full).contiguous(), None, None


def reduce_visibility_(visible: torch.Tensor, part: RowPartition, group=None) -> torch.Tensor:
    """In-place union of the per-band visibility flags (uint8/bool, MAX)."""
    if part.world_size > 1:
        buf = visible if visible.dtype == torch.uint8 else visible.to(torch.uint8)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        if buf is not visible:
            visible.copy_(buf.to(visible.dtype))
    return visible


def reduce_grads_(*grads: torch.Tensor, part: RowPartition, group=None):
    """In-place SUM of per-band gradient partials; small tensors are flattened into one bucket so a
    step issues a single all-reduce."""
    if part.world_size == 1 or not grads:
        return grads
    if len(grads) == 1:
        dist.all_reduce(grads[0], op=dist.ReduceOp.SUM, group=group)
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return grads


class _BandImageLoss(torch.autograd.Function):
    """Image loss of a row band with the per-image sums all-reduced over the ranks: every rank gets the GLOBAL loss
    value and the gradient of it with respect to its own band.  Two launches + one collective: block partials of the band
    (`ops.image_loss_band_partials`), all-reduce (SUM) of the partials (20 KB at 8 cameras), band gradient + the losses in
    one launch (`ops.image_loss_band_backward_partials`, kept for the backward: the loss is linear in `grad_total`)."""

    @staticmethod
    def forward(ctx, rgba_band, img, mask_img, rows, lambda_rgb, lambda_silhouette, group):
        from . import ops
        part = ops.image_loss_band_partials(rgba_band, img, mask_img, rows)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
        grad, losses = ops.image_loss_band_backward_partials(rgba_band, img, mask_img, rows, lambda_rgb, lambda_silhouette, part)
        ctx.save_for_backward(grad)
        total, rest = losses[0], losses[1:]
        ctx.mark_non_differentiable(rest)
        return total, rest

    @staticmethod
    def backward(ctx, grad_total, _grad_rest):
        (grad,) = ctx.saved_tensors
        return grad * grad_total, None, None, None, None, None, None


def band_image_loss(rgba_band, img, mask_img, part: RowPartition, lambda_dr_rgb: float = 1.0,
                    lambda_dr_silhouette: float = 1.0, group=None):
    """``Trainer.calc_dr_loss`` (trainer.py:332-372) for a row-partitioned render: ``rgba_band`` (N, rows, S, 4) is this
    rank's band, ``img`` (N,H,W,3) / ``mask_img`` (N,H,W) the full targets (replicated).  The masked-mean denominators
    and the IoU intersections / unions are global sums: one all-reduce of 5 doubles per image; the backward
    then only needs the band.  Returns the same dictionary as `dss_amd.losses.calc_dr_loss`, identical on every rank."""
    if mask_img.dtype != torch.float32:
        mask_img = mask_img.float()
    total, rest = _BandImageLoss.apply(rgba_band, img, mask_img, part.rows, float(lambda_dr_rgb),
                                       float(lambda_dr_silhouette), group)
    return {"loss": total, "loss_dr_rgb": rest[0], "loss_dr_silhouette": rest[1], "loss_iou": rest[2]}

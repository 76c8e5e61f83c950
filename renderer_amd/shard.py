"""Work split across GPUs: sort-first screen bands and frame-parallel views, plus the band gather.

Nothing in the reference is distributed (one queue, one submission per frame,
src/renderer/drawer.rs:426-454).  The split follows the two ways its frame decomposes:

* sort-first: the framebuffer is cut into horizontal bands of whole tile rows, one band per rank;
  geometry is replicated, bands are pixel-disjoint, so the only exchange is the final gather of
  the finished RGBA8 rows (grouped send/receive: RCCL over xGMI when the backend is "nccl", gloo
  on CPU), overlapped with the drawing of the next frame (SortFirstPipeline); exchange_bands leaves
  the whole frame on every rank instead;
* frame-parallel: independent views are dealt to ranks in contiguous blocks; no exchange at all.
"""
import numpy as np

TILE = 32


def _tile_rows_per_rank(n_tile_rows, world, weights):
    """Whole tile rows per rank in proportion to the weights (largest remainders first; a rank with
    a positive weight keeps at least one row while there are rows to go round)."""
    w = [max(float(x), 0.0) for x in weights]
    if len(w) != world or not sum(w) > 0.0:
        raise ValueError("weights must be %d non-negative numbers with a positive sum" % world)
    ideal = [n_tile_rows * x / sum(w) for x in w]
    rows = [int(x) for x in ideal]
    by_remainder = sorted(range(world), key=lambda r: (-(ideal[r] - rows[r]), r))
    for r in by_remainder[:n_tile_rows - sum(rows)]:
        rows[r] += 1
    for r in range(world):  # nobody with work to do is left without a row
        if w[r] > 0.0 and rows[r] == 0 and max(rows) > 1:
            rows[rows.index(max(rows))] -= 1
            rows[r] = 1
    return rows


def band_rows(height, world, rank, tile=TILE, weights=None):
    """Rows [y0, y1) of rank's band: contiguous, whole tile rows.  Without weights the bands differ
    by at most one tile row; with weights (one number per rank) their heights are in that proportion."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %r of %r" % (rank, world))
    n_tile_rows = (height + tile - 1) // tile
    if weights is None:
        base, extra = divmod(n_tile_rows, world)
        first = rank * base + min(rank, extra)
        count = base + (1 if rank < extra else 0)
    else:
        rows = _tile_rows_per_rank(n_tile_rows, world, weights)
        first, count = sum(rows[:rank]), rows[rank]
    y0 = min(first * tile, height)
    y1 = min((first + count) * tile, height)
    return y0, y1


def all_bands(height, world, tile=TILE, weights=None):
    return [band_rows(height, world, r, tile, weights) for r in range(world)]


def balanced_weights(world, dst, fixed_ms, band_ms, link_ms):
    """Band weights for a sort-first split whose frames are gathered on `dst` while the next ones are
    drawn.  A rank with share x of the rows needs fixed_ms + band_ms * x to draw its band (the
    geometry stage is replicated, the tile stage follows the rows) and, unless it is `dst`,
    link_ms * x to send it, every sender over a link of its own.  With equal bands and few ranks
    the senders' links set the pace (two ranks: half a frame over one link); giving `dst`, which
    sends nothing, a taller band s evens that out: fixed + band * s = link * (1 - s) / (world - 1).
    Equal weights when drawing is the slower side anyway."""
    even = [1.0] * world
    if world < 2 or not (link_ms > 0.0 and band_ms >= 0.0 and fixed_ms >= 0.0):
        return even
    if link_ms / world <= fixed_ms + band_ms / world:
        return even
    per_sender = link_ms / (world - 1)
    s = (per_sender - fixed_ms) / (band_ms + per_sender)
    s = min(max(s, 1.0 / world), 0.9)
    return [s if r == dst else (1.0 - s) / (world - 1) for r in range(world)]


def views_for_rank(n_views, world, rank):
    """Contiguous block [v0, v1) of the views rendered by this rank."""
    base, extra = divmod(n_views, world)
    v0 = rank * base + min(rank, extra)
    return v0, v0 + base + (1 if rank < extra else 0)


def gather_bands(band, height, width, world, rank, dst=0, group=None, out=None, weights=None):
    """Collect every rank's finished band on rank `dst`: one send per rank, one receive per sender
    on `dst`, all posted as one group (on RCCL the receives then run concurrently, one xGMI link
    per sending GPU), every band at its own size, straight into its rows of the frame.

    `band` is this rank's rows as a (rows, width, 4) uint8 torch tensor (a CUDA tensor for the
    nccl/RCCL backend, a CPU tensor for gloo).  On `dst` the frame is assembled in `out` (a
    (height, width, 4) uint8 tensor on the band's device; allocated when None) and returned;
    other ranks return None.  Ranks whose band is empty take no part in the exchange.  `weights`:
    the band heights' proportions, as for band_rows (every rank must pass the same).
    CUDA tensors on a gloo group (a rehearsal of several ranks on fewer GPUs) go through host
    memory: gloo does not send from device memory.
    """
    import torch
    import torch.distributed as dist

    bands = all_bands(height, world, weights=weights)
    y0, y1 = bands[rank]
    if tuple(band.shape) != (y1 - y0, width, 4) or band.dtype != torch.uint8:
        raise ValueError("band has shape %s, expected %s uint8" % (tuple(band.shape), (y1 - y0, width, 4)))
    band = band.contiguous()
    via_host = band.is_cuda and world > 1 and dist.get_backend(group) == "gloo"
    if rank != dst:
        if y1 > y0 and world > 1:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, band.cpu() if via_host else band, dst, group)]):
                w.wait()
        return None
    frame = out if out is not None else torch.empty((height, width, 4), dtype=torch.uint8, device=band.device)
    if tuple(frame.shape) != (height, width, 4) or frame.dtype != torch.uint8 or not frame.is_contiguous():
        raise ValueError("out must be a contiguous (%d, %d, 4) uint8 tensor" % (height, width))
    senders = [(r, a, b) for r, (a, b) in enumerate(bands) if r != dst and b > a]
    landing = {r: (torch.empty((b - a, width, 4), dtype=torch.uint8) if via_host else frame[a:b]) for r, a, b in senders}
    ops = [dist.P2POp(dist.irecv, landing[r], r, group) for r, a, b in senders]
    if y1 > y0:
        frame[y0:y1].copy_(band)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if via_host:
        for r, a, b in senders:
            frame[a:b].copy_(landing[r])
    return frame


def exchange_bands(band, height, width, world, rank, group=None, out=None, weights=None):
    """Every rank ends up with the whole frame (a consumer on each GPU: a post-process, an encoder per
    GPU): each rank sends its band to every other rank and receives theirs, all as one group of
    point-to-point operations, every band at its own size, straight into its rows of `out`.
    gather_bands is the cheaper exchange when one rank presents the frame; this one moves
    (world - 1) times as many bytes, every rank's own band over each of its links.
    Arguments as for gather_bands; returns the assembled (height, width, 4) uint8 tensor on every rank."""
    import torch
    import torch.distributed as dist

    bands = all_bands(height, world, weights=weights)
    y0, y1 = bands[rank]
    if tuple(band.shape) != (y1 - y0, width, 4) or band.dtype != torch.uint8:
        raise ValueError("band has shape %s, expected %s uint8" % (tuple(band.shape), (y1 - y0, width, 4)))
    band = band.contiguous()
    frame = out if out is not None else torch.empty((height, width, 4), dtype=torch.uint8, device=band.device)
    if tuple(frame.shape) != (height, width, 4) or frame.dtype != torch.uint8 or not frame.is_contiguous():
        raise ValueError("out must be a contiguous (%d, %d, 4) uint8 tensor" % (height, width))
    if y1 > y0:
        frame[y0:y1].copy_(band)
    if world == 1:
        return frame
    via_host = band.is_cuda and dist.get_backend(group) == "gloo"
    others = [(r, a, b) for r, (a, b) in enumerate(bands) if r != rank and b > a]
    landing = {r: (torch.empty((b - a, width, 4), dtype=torch.uint8) if via_host else frame[a:b]) for r, a, b in others}
    mine = band.cpu() if via_host else band
    ops = [dist.P2POp(dist.irecv, landing[r], r, group) for r, a, b in others]
    if y1 > y0:
        ops += [dist.P2POp(dist.isend, mine, r, group) for r in range(world) if r != rank]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if via_host:
        for r, a, b in others:
            frame[a:b].copy_(landing[r])
    return frame


class _DevicePtr:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr="|u1"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def band_tensor(drawer, image_index, y0, y1):
    """The rows [y0,y1) of a drawer's colour image as a CUDA uint8 tensor sharing its memory."""
    import torch

    ptr = drawer.color_device_ptr(image_index)
    full = torch.as_tensor(_DevicePtr(ptr, (drawer.height, drawer.width, 4)), device="cuda:%d" % drawer.cfg.device)
    return full[y0:y1]


def frame_hash(color):
    """Order-sensitive 64-bit checksum of an RGBA8 frame (numpy), for cross-rank equality checks."""
    a = np.ascontiguousarray(color).view(np.uint8).ravel().astype(np.uint64)
    idx = np.arange(1, a.size + 1, dtype=np.uint64)
    return int(((a + np.uint64(1)) * (idx * np.uint64(0x9E3779B97F4A7C15) | np.uint64(1))).sum(dtype=np.uint64))


class StaleBand(RuntimeError):
    """Frames were drawn again after their bands had been gathered: the assembled frames are not final."""


class SortFirstPipeline:
    """Sort-first rendering of a frame sequence on one rank: this rank's band of frame i is drawn
    into image i % 2 on the drawer's own stream while the band of frame i - 1 is being gathered on
    a second stream, ordered by events only (the host never waits inside the loop).

    The drawer must have been created with this rank's scissor band (band_rows with the same
    `weights`) and at least two images.
    A frame that overflows a work buffer is drawn again by the library when it takes the frame's
    result, which is after its band was sent: the band that left was incomplete.  It cannot happen
    once the buffers have grown to the scene (draw a frame and `sync()` before pipelining), and when
    it does happen it is not silent: `finish()` raises `StaleBand` (with `strict=False` it records
    the count in `stale_frames` instead), and the caller submits those frames again.
    """

    def __init__(self, drawer, height, width, world, rank, dst=0, group=None, weights=None, strict=True):
        import torch
        self.strict, self.stale_frames = strict, 0
        self._redraws_seen = drawer.stats()["redraws"]
        self.d, self.h, self.w, self.world, self.rank, self.dst, self.group = drawer, height, width, world, rank, dst, group
        self.weights = weights
        self.y0, self.y1 = band_rows(height, world, rank, weights=weights)
        dev = torch.device("cuda", drawer.cfg.device)
        self.lib_stream = torch.cuda.ExternalStream(drawer.stream_ptr(), device=dev)
        self.side = torch.cuda.Stream(device=dev)
        self.drawn = [torch.cuda.Event() for _ in range(2)]
        self.gathered = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.gather_start = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.used = [False, False]
        self.frame = torch.empty((height, width, 4), dtype=torch.uint8, device=dev) if rank == dst else None
        self.bands = [band_tensor(drawer, img, self.y0, self.y1) for img in range(2)]

    def submit(self, i):
        """Draw frame i (the caller has set uniforms and camera) and start gathering it."""
        import torch
        img = i & 1
        if self.used[img]:
            self.lib_stream.wait_event(self.gathered[img])  # the band of frame i - 2 has left this image
        self.d.draw(img)
        self.drawn[img].record(self.lib_stream)
        self.side.wait_event(self.drawn[img])
        with torch.cuda.stream(self.side):
            self.gather_start[img].record(self.side)
            gather_bands(self.bands[img], self.h, self.w, self.world, self.rank, self.dst, self.group, out=self.frame, weights=self.weights)
            self.gathered[img].record(self.side)
        self.used[img] = True

    def finish(self):
        """Wait for everything submitted; returns the last frame on `dst` (else None)."""
        self.d.sync()
        self.side.synchronize()
        redraws = self.d.stats()["redraws"]
        if redraws != self._redraws_seen:
            # the library drew frames again (after growing a work buffer) when it took their results:
            # by then their bands had been sent as they were
            stale, self._redraws_seen = redraws - self._redraws_seen, redraws
            self.stale_frames += stale
            if self.strict:
                raise StaleBand("%d frame(s) were drawn again after their bands had left rank %d: draw one frame and sync() "
                                "before pipelining, or submit the frames again" % (stale, self.rank))
        return self.frame

    def last_gather_ms(self, i):
        """Device time of frame i's gather on this rank (after finish())."""
        return self.gather_start[i & 1].elapsed_time(self.gathered[i & 1])


class ViewPool:
    """Many small views of one scene (the multi-view configuration) on one GPU.

    A 512x512 view of a hundred objects keeps the GPU busy for a few microseconds per kernel, so
    a single stream spends most of a view waiting for launches.  The pool holds `contexts`
    independent drawers (own stream, own work buffers, the scene uploaded to each) and deals the
    views out over them in turn, so that the kernels of several views are in flight at once.
    Nothing is shared between the contexts and no kernel knows about the others; every view is
    rendered exactly as a single drawer would render it.
    """

    def __init__(self, scene, contexts=4, n_images=2, device=0, **drawer_args):
        from . import api
        self.size = scene.size
        self.n_images = n_images
        self.drawers = []
        for _ in range(max(1, int(contexts))):
            d = api.HipDrawer(scene.size[0], scene.size[1], device=device, n_images=n_images, **drawer_args)
            d.update_scene(scene)
            self.drawers.append(d)

    def close(self):
        for d in self.drawers:
            d.close()
        self.drawers = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render(self, cameras, read=None):
        """Render one view per camera Transform.  `read(drawer, image_index)` turns a finished view
        into what is returned for it (default: its RGBA8 image); it is called when the view's image
        is about to be used again and for the rest at the end, always in view order per context."""
        if read is None:
            read = lambda d, img: d.read_color(img)  # noqa: E731
        n = len(cameras)
        out = [None] * n
        k_ctx = len(self.drawers)
        holds = [[None] * self.n_images for _ in range(k_ctx)]  # which view each image holds, unread
        for i in range(n):
            c = i % k_ctx
            img = (i // k_ctx) % self.n_images
            d = self.drawers[c]
            if holds[c][img] is not None:
                out[holds[c][img]] = read(d, img)
            d.update_camera(cameras[i:i + 1])
            d.draw(img)
            holds[c][img] = i
        for c in range(k_ctx):
            for v, img in sorted((v, img) for img, v in enumerate(holds[c]) if v is not None):
                out[v] = read(self.drawers[c], img)
        return out

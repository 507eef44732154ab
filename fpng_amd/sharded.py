"""Multi-GPU sharding of the encode path: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two regimes:

* batches of images -- images are independent objects: each rank encodes its own slice, there is NO
  data-path collective.  `gather_pngs()` optionally brings the finished files to one rank
  (all_gather of sizes + padded payload).

* one huge image -- rows are sharded into contiguous bands, but the output stays ONE IDAT / ONE
  zlib stream / ONE Deflate block like the reference (reference fpng.cpp:1764-1800; the fpng decoder
  rejects a second IDAT, fpng.cpp:3032-3033).  Bands therefore meet at BIT granularity:
    0. (2-pass only) every rank histograms its band; one all_reduce of 288 counters; every rank builds
       the same Huffman table from the sum
    1. every rank encodes its band into scratch streams and learns its token bits, Adler partial sums,
       last flush unit
    2. all_gather of one small record per rank (the only exchange on the critical path)
    3. everyone derives every band's start bit (exclusive prefix sum), the global Adler-32 and the
       reference's "ran out of buffer -> stored blocks" decision (closed form)
    4. every rank shifts its band to its bit position inside a WINDOW of whole 16-byte pieces of the
       file (foreign bits 0)
    5. windows go to the root, which receives them straight into the file buffer (its own band is
       placed there directly); neighbouring windows share at most one 16-byte piece: a window's first
       piece travels on its own when it is shared and is OR-ed onto the predecessor's last piece
    6. the IDAT CRC-32 is sharded too: the per-64-KiB partials each rank's placement kernel computes
       XOR together (a raw CRC is linear, windows are zero in foreign bits); one gather of a few KiB
    7. the root wraps the stream: PNG header, Adler-32, CRC fold, IEND -- without reading the file.

The arithmetic of steps 2-5 is plain Python here; the per-band work is done by a "band backend":
the HIP encoder on GPUs (`GpuBandBackend`), or a CPU stand-in injected by the gloo tests.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist

ADLER_MOD = 65521
ENCODE_SLOWER = 1


@dataclass
class BandStats:
    token_bits: int
    s1: int          # raw byte sum of the band's filtered bytes, mod 65521
    s2: int          # raw position-weighted sum, mod 65521
    nbytes: int      # filtered bytes in the band
    last_unit_bits: int
    first_token_bit: int = 0   # same on every rank (1-pass: constant; 2-pass: length of the dynamic header)
    eob_bits: int = 12


@dataclass
class BandPlan:
    start_bits: list     # absolute zlib bit where each band's tokens start
    end_bit: int         # zlib bit after the last token
    adler: int           # Adler-32 of the whole filtered stream
    stored: bool         # reference would have fallen back to stored blocks
    zlib_size: int       # bytes of the compressed zlib stream (incl. Adler) when not stored


def split_rows(h, world):
    """Contiguous row bands, as even as possible; bands may be empty only if h < world."""
    base, rem = divmod(h, world)
    cuts = [0]
    for r in range(world):
        cuts.append(cuts[-1] + base + (1 if r < rem else 0))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def plan_bands(stats, w, h, c, first_token_bit, eob_bits, one_pass=True):
    """Steps 2-3: from the gathered per-band records to the global layout."""
    start_bits, pos = [], first_token_bit
    for s in stats:
        start_bits.append(pos)
        pos += s.token_bits
    end_bit = pos
    # Adler-32 of X||Y from raw sums: S1 adds, S2(XY) = S2(X) + |Y|*S1(X) + S2(Y)   
    s1 = s2 = n = 0
    for s in stats:
        s2 = (s2 + (s.nbytes % ADLER_MOD) * s1 + s.s2) % ADLER_MOD  # s1 is still S1 of everything before
        s1 = (s1 + s.s1) % ADLER_MOD
        n += s.nbytes
    adler = (((n % ADLER_MOD + s2) % ADLER_MOD) << 16) | ((1 + s1) % ADLER_MOD)
    # failure rule of the reference's bit writer (reference fpng.cpp:567-588), closed form
    n_total = (w * c + 1) * h
    D = ((58 + n_total + 7) & ~7) - 58
    last_unit = 0
    for s in stats:
        if s.nbytes:
            last_unit = s.last_unit_bits
    stored = (one_pass and D < first_token_bit // 8) or (((end_bit - last_unit) >> 3) + 8 > D) or (((end_bit + eob_bits + 7) >> 3) + 4 > D)
    zlib_size = ((end_bit + eob_bits + 7) >> 3) + 4
    return BandPlan(start_bits, end_bit, adler, stored, zlib_size)


class GpuBandBackend:
    """Per-band work on this rank's GPU through the C ABI (fpng_amd_band_hist / _encode / _place / _crc_partials / _wrap_png)."""
    has_crc_partials = True

    def __init__(self, encoder):
        # the band calls are asynchronous on the encoder's stream and their results (windows, histograms, CRC partials) are
        # consumed by torch ops and collectives: both must be the same stream
        if not getattr(encoder, "_follow_torch", False):
            raise ValueError('GpuBandBackend needs an Encoder(stream="torch"): its outputs are consumed on torch\'s current stream')
        self.enc = encoder

    def hist(self, rows, row_above, w, c, y0, y1, h):
        hist = torch.empty(288, dtype=torch.int32, device=rows.device)
        self.enc.band_hist(self.enc._band(rows, row_above, w, c, y0, y1, h), hist)
        return hist

    def encode(self, rows, row_above, w, c, y0, y1, h, flags, hist):
        self._band = self.enc._band(rows, row_above, w, c, y0, y1, h)
        self._keep = (rows, row_above, hist)
        st = self.enc.band_encode(self._band, flags, hist)
        return BandStats(st.token_bits, st.adler_s1, st.adler_s2, st.adler_len, st.last_unit_bits, st.first_token_bit, st.eob_bits)

    def place(self, start_bit, zlib_size, token_bits, device, out=None):
        """-> (file offset of the window, window tensor): 16-byte pieces of the file, foreign bits zero.
        out: storage for the window (16-byte aligned uint8 tensor, e.g. the file buffer from window_offset() on)."""
        if out is None:
            cap = ((token_bits + 7) >> 3) + 64 + 512
            out = torch.empty((cap + 15) & ~15, dtype=torch.uint8, device=device)
        off, n = self.enc.band_place(self._band, start_bit, zlib_size, out)
        return off, out[:n]

    def crc_partials(self, device):
        """The placed band's contribution to the IDAT CRC: one int32 per 64 KiB range of the file (XOR over the bands =
        the file's partials; a raw CRC is linear and a window is zero where other bands' bits are)."""
        return self.enc.band_crc_partials(device)

    def wrap(self, png_buf, zlib_size, adler, w, h, c, crc_partials=None):
        n = self.enc.wrap_png(png_buf, zlib_size, adler, w, h, c, crc_partials)
        return png_buf[:n]

    def encode_whole(self, image, w, h, c, flags):
        pngs, _ = self.enc.encode_tensors([image], flags)
        return pngs[0]


def window_offset(first, start_bit):
    """File offset of a band's window: the 16-byte piece holding its first token bit (0 for the image's first band, whose
    window also carries the stream's head) -- the rule of fpng_amd_band_place()."""
    return 0 if first else ((58 * 8 + start_bit) >> 3) & ~15


def window_extent(first, last, start_bit, token_bits, eob_bits):
    """(file offset, bytes) of a band's window -- every rank can work out every band's from the gathered records."""
    fb0 = 58 * 8 + start_bit
    fb1 = fb0 + token_bits + (eob_bits if last else 0)
    wb0 = window_offset(first, start_bit)
    return wb0, ((((fb1 + 7) >> 3) + 15) & ~15) - wb0


def merge_window(png_buf, file_off, win, first, start_bit):
    """Step 5: a band's window into the file.  Everything is copied except the window's first 16-byte piece when the
    band shares it with its predecessor (each wrote zeros where the other's bits are): that one is OR-ed.  A band that
    starts exactly on a piece boundary shares nothing."""
    n = win.numel()
    if first:
        png_buf[58:n] = win[58:]   # (the PNG header's bytes of the first window are undefined: wrap() writes them)
        return
    head = min(16, n) if (58 * 8 + start_bit) % 128 else 0
    view = png_buf[file_off:file_off + head]
    torch.bitwise_or(view, win[:head], out=view)
    if n > head:
        png_buf[file_off + head:file_off + n] = win[head:]


def encode_image_bands_local(backend, image, cuts, flags=0):
    """Single-process version of the band pipeline (bands processed one after another on one GPU):
    the same hist -> encode -> plan -> place -> merge -> wrap steps, used by the GPU parity tests and handy
    for images too tall for one submission.  image: uint8 tensor (h, w, c); cuts: row boundaries."""
    h, w, c = image.shape
    bands = [(y0, y1) for y0, y1 in zip(cuts[:-1], cuts[1:]) if y1 > y0]
    above = lambda y0: image[y0 - 1] if y0 else None  # noqa: E731
    hist = None
    if flags & ENCODE_SLOWER:
        hist = sum(backend.hist(image[y0:y1], above(y0), w, c, y0, y1, h).to(torch.int64) for y0, y1 in bands).to(torch.int32)
    # one encoder holds one band's streams at a time: encode and place band after band (two walks over the rows
    # in this single-GPU form; on N GPUs every rank keeps its band's streams between the two phases)
    stats = [backend.encode(image[y0:y1], above(y0), w, c, y0, y1, h, flags, hist) for y0, y1 in bands]
    plan = plan_bands(stats, w, h, c, stats[0].first_token_bit, stats[0].eob_bits, not (flags & ENCODE_SLOWER))
    if plan.stored:
        return backend.encode_whole(image, w, h, c, 2)
    png_buf = torch.empty(((58 + plan.zlib_size + 16 + 15) & ~15) + 16, dtype=torch.uint8, device=image.device)
    crc = None
    for i, (y0, y1) in enumerate(bands):
        backend.encode(image[y0:y1], above(y0), w, c, y0, y1, h, flags, hist)
        off, win = backend.place(plan.start_bits[i], plan.zlib_size, stats[i].token_bits, image.device)
        merge_window(png_buf, off, win, i == 0, plan.start_bits[i])
        part = backend.crc_partials(image.device)  # None: this backend leaves the CRC to wrap()
        if part is not None:
            crc = part if crc is None else torch.bitwise_xor(crc, part)
    return bytes(backend.wrap(png_buf, plan.zlib_size, plan.adler, w, h, c, crc).cpu().numpy())


def _all_gather_records(rec, group, device):
    world = dist.get_world_size(group)
    t = torch.tensor(rec, dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [o.tolist() for o in out]


def encode_image_row_sharded(backend, rows, row_above, w, h, c, y0, y1, flags=0, group=None, root=0):
    """Encode ONE w x h image whose rows [y0,y1) live on this rank (`rows`: uint8 tensor (y1-y0, w, c);
    `row_above`: the image row y0-1 (tensor (w, c)) or None when y0 == 0).  flags: 0 or FPNG_ENCODE_SLOWER.
    Returns the PNG as a uint8 tensor on the root, None elsewhere.  Output is byte-identical to the
    single-GPU / reference encoding of the whole image."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = rows.device
    nrows = y1 - y0
    hist = None
    if flags & ENCODE_SLOWER:
        # step 0: the image's histogram = sum of the bands' (288 counters through one all_reduce)
        hist = backend.hist(rows, row_above, w, c, y0, y1, h) if nrows > 0 else torch.zeros(288, dtype=torch.int32, device=device)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    st = backend.encode(rows, row_above, w, c, y0, y1, h, flags, hist) if nrows > 0 else BandStats(0, 0, 0, 0, 0, 0, 0)
    recs = _all_gather_records([st.token_bits, st.s1, st.s2, st.nbytes, st.last_unit_bits, st.first_token_bit, st.eob_bits, y0, y1],
                               group, device)
    order = sorted(range(world), key=lambda r: (recs[r][7], recs[r][8]))  # bands in row order
    stats = [BandStats(*recs[r][:7]) for r in order]
    non_empty = [r for r in order if recs[r][8] > recs[r][7]]
    ftb, eob = recs[non_empty[0]][5], recs[non_empty[0]][6]
    plan = plan_bands(stats, w, h, c, ftb, eob, not (flags & ENCODE_SLOWER))
    my_pos = order.index(rank)

    if plan.stored:
        # Rare path (incompressible image): the stored-block layout has no bit seams; the root collects the raw
        # rows (padded all_gather of equal-size tensors) and writes the stored stream.
        max_rows = max(recs[r][8] - recs[r][7] for r in range(world))
        pad = torch.zeros((max_rows, w, c), dtype=torch.uint8, device=device)
        if nrows:
            pad[:nrows] = rows
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        if rank != root:
            return None
        img = torch.cat([parts[r][:recs[r][8] - recs[r][7]] for r in order]).contiguous()
        png = backend.encode_whole(img, w, h, c, 2)
        import numpy as np
        return torch.from_numpy(np.frombuffer(png, dtype=np.uint8).copy())

    win, off, crc, png_buf = None, 0, None, None
    if rank == root:
        png_buf = torch.empty(((58 + plan.zlib_size + 16 + 15) & ~15) + 16, dtype=torch.uint8, device=device)
    if nrows > 0:
        # the root places its own band straight into the file buffer; the others into a window that is sent over
        dest = None
        if rank == root:
            dest = png_buf[window_offset(rank == non_empty[0], plan.start_bits[my_pos]):]
        off, win = backend.place(plan.start_bits[my_pos], plan.zlib_size, st.token_bits, device, out=dest)
        crc = backend.crc_partials(device)
    # the IDAT CRC, sharded like the rows: every rank's per-range partials (a few KiB), XOR-ed on the root.  Whether the
    # backend produces them is a property of its class (the same on every rank); their number follows from zlib_size.
    crc_all = None
    if getattr(backend, "has_crc_partials", False):
        n_part = ((((58 + plan.zlib_size - 4) + 15) & ~15) - 48 + 65535) >> 16
        assert crc is None or int(crc.numel()) == n_part
        mine = crc if crc is not None else torch.zeros(n_part, dtype=torch.int32, device=device)
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == root else None
        dist.gather(mine, parts, dst=_global_rank(group, root), group=group)
        if rank == root:
            crc_all = parts[0]
            for p_ in parts[1:]:
                crc_all = torch.bitwise_xor(crc_all, p_)

    # ---- step 5: windows to the root; the offsets follow from the plan, the sizes are exchanged.  A remote window is
    #      received straight into its place in the file, except its first 16-byte piece when the band shares that piece with
    #      its predecessor (both wrote zeros where the other's bits are): that piece travels on its own and is OR-ed in ----
    geo = [None] * world  # (file offset, bytes) of every band's window: from the plan, no exchange
    for pos, r in enumerate(order):
        geo[r] = (0, 0)
        if recs[r][8] > recs[r][7]:
            geo[r] = window_extent(r == non_empty[0], recs[r][8] == h, plan.start_bits[pos], recs[r][0], eob)
    assert win is None or geo[rank] == (off, int(win.numel())), (geo[rank], off, win.numel())

    def shared_head(r):  # bytes of band r's window that overlap its predecessor's window
        if r == non_empty[0]:
            return 0
        return min(16, geo[r][1]) if (58 * 8 + plan.start_bits[order.index(r)]) % 128 else 0

    if rank == root:
        pending, heads = [], {}
        if win is not None:  # the root's own band is in place already; a shared first piece is set aside: the predecessor's
            hd = shared_head(root)  # window (received below) overwrites it, then it is OR-ed back in
            if hd:
                heads[root] = win[:hd].clone()
        for r in non_empty:
            if r == root:
                continue
            o, n, hd = geo[r][0], geo[r][1], shared_head(r)
            if hd:
                heads[r] = torch.empty(hd, dtype=torch.uint8, device=device)
                pending.append(dist.irecv(heads[r], src=_global_rank(group, r), group=group))
            if n > hd:
                pending.append(dist.irecv(png_buf[o + hd:o + n], src=_global_rank(group, r), group=group))
        for req in pending:
            req.wait()
        for r in non_empty:  # the shared pieces: OR-ed onto what the predecessor's window put there
            hd = shared_head(r)
            if hd:
                view = png_buf[geo[r][0]:geo[r][0] + hd]
                torch.bitwise_or(view, heads[r], out=view)
        return backend.wrap(png_buf, plan.zlib_size, plan.adler, w, h, c, crc_all)
    if win is not None:
        hd = shared_head(rank)
        win = win.contiguous()
        if hd:
            dist.send(win[:hd].clone(), dst=_global_rank(group, root), group=group)
        if win.numel() > hd:
            dist.send(win[hd:], dst=_global_rank(group, root), group=group)
    return None


class CppRowSharded:
    """The same exchange behind the C ABI (fpng_amd_encode_image_sharded, fpng_amd/csrc/sharded.cpp) over the built-in RCCL
    transport -- what a C++ host uses; no torch.distributed on the data path.  The 128-byte RCCL id is made on rank 0
    (`rccl_unique_id()`) and must reach every rank before construction (here: any torch.distributed broadcast)."""

    def __init__(self, encoder, rank, world, unique_id, device):
        import ctypes as C
        from . import _lib
        self.enc, self.rank, self.world = encoder, rank, world
        self.lib = _lib.load()
        self.t = C.POINTER(_lib.Transport)()
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(self.lib.fpng_amd_rccl_transport_create(C.byref(self.t), idb, rank, world, device))

    @staticmethod
    def rccl_unique_id():
        import ctypes as C
        from . import _lib
        b = (C.c_uint8 * 128)()
        _lib.check(_lib.load().fpng_amd_rccl_unique_id(b))
        return bytes(b)

    def encode(self, rows, row_above, w, h, c, y0, y1, flags=0, root=0, out=None):
        """rows: uint8 CUDA tensor (y1-y0, w, c) (may be empty), out: uint8 CUDA tensor of >= max_encoded_size + 64 bytes on the
        root.  Returns the PNG as a view of `out` on the root, None elsewhere."""
        import ctypes as C
        from . import _lib
        self.enc._sync_stream()
        b = _lib.Band()
        b.d_rows = rows.data_ptr() if y1 > y0 else None
        b.d_row_above = row_above.data_ptr() if (row_above is not None and y0 > 0) else None
        b.w, b.num_chans, b.y0, b.y1, b.h_total = w, c, y0, y1, h
        n = C.c_size_t(0)
        _lib.check(self.lib.fpng_amd_encode_image_sharded(self.enc.h, self.t, C.byref(b), flags, root,
                                                          out.data_ptr() if out is not None else None, out.numel() if out is not None else 0, C.byref(n)))
        return out[: n.value] if self.rank == root else None

    def close(self):
        if self.t:
            self.lib.fpng_amd_rccl_transport_destroy(self.t)
            self.t = None


def _global_rank(group, r):
    return r if group is None else dist.get_global_rank(group, r)


def shard_batch(n_images, rank, world):
    """Contiguous block of a batch for this rank (images are independent: no collective)."""
    lo, hi = split_rows(n_images, world)[rank]
    return range(lo, hi)


def gather_pngs(pngs, group=None, root=0, device=None):
    """Optional: bring every rank's finished PNG files to `root` (variable sizes -> all_gather of the
    sizes, then one padded all_gather of the payload).  Returns list-of-bytes on root, else None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
    import numpy as np
    sizes = [len(p) for p in pngs]
    counts = _all_gather_records([len(sizes)], group, device)
    max_n = max(c[0] for c in counts)
    all_sizes = _all_gather_records(sizes + [0] * (max_n - len(sizes)), group, device) if max_n else [[] for _ in range(world)]
    max_bytes = max([sum(s) for s in all_sizes] + [1])
    payload = torch.zeros(max_bytes, dtype=torch.uint8, device=device)
    blob = b"".join(pngs)
    if blob:
        payload[:len(blob)] = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    if rank != root:
        return None
    out = []
    for r in range(world):
        data = gathered[r].cpu().numpy().tobytes()
        off = 0
        for s in all_sizes[r][:counts[r][0]]:
            out.append(data[off:off + s])
            off += s
    return out

"""Host-side mirror of the reference's `namespace fpng` encode interface (reference src/fpng.h:17-52)
on top of the C ABI, plus the device-resident batch / row-band entry points the MI355X path adds.

Names, argument meaning and error behaviour follow the reference:

    fpng_init()                                           src/fpng.h:17
    fpng_cpu_supports_sse41()  -> "is the accelerator usable"   src/fpng.h:23
    fpng_crc32(data, prev=0) / fpng_adler32(data, prev=1)        src/fpng.h:26-31
    fpng_encode_image_to_memory(image, w, h, num_chans, flags) -> (ok, bytes)   src/fpng.h:48
    fpng_encode_image_to_file(filename, image, w, h, num_chans, flags) -> ok    src/fpng.h:52

torch is only plumbing here (device memory + streams).  There is no CPU fallback: if the HIP library
or a GPU is missing, encode calls raise.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Band, BandStats, FpngAmdError, HostImage, Image, Result, check

FPNG_ENCODE_SLOWER = 1        # reference src/fpng.h:38
FPNG_FORCE_UNCOMPRESSED = 2   # reference src/fpng.h:41
FPNG_CRC32_INIT = 0           # reference src/fpng.h:26
FPNG_ADLER32_INIT = 1         # reference src/fpng.h:30

MODE_COMPRESSED, MODE_STORED = 0, 1
SYNTH_KINDS = {"noise": 0, "solid": 1, "grad": 2, "blocks": 3}


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def fpng_init(device=-1):
    check(_lib.load().fpng_amd_init(device))


def fpng_cpu_supports_sse41():
    """Kept for source compatibility; answers "is the MI355X path usable"."""
    return bool(_lib.load().fpng_amd_device_available())


def fpng_crc32(data, prev_crc32=FPNG_CRC32_INIT):
    b = _as_u8(data)
    return _lib.load().fpng_amd_crc32(b.ctypes.data, b.size, prev_crc32)


def fpng_adler32(data, adler=FPNG_ADLER32_INIT):
    b = _as_u8(data)
    return _lib.load().fpng_amd_adler32(b.ctypes.data, b.size, adler)


def crc32_combine(crc_x, crc_y, len_y):
    return _lib.load().fpng_amd_crc32_combine(crc_x, crc_y, len_y)


def adler32_combine(adler_x, adler_y, len_y):
    return _lib.load().fpng_amd_adler32_combine(adler_x, adler_y, len_y)


def max_encoded_size(w, h, num_chans):
    return _lib.load().fpng_amd_max_encoded_size(w, h, num_chans)


def layout_1pass(num_chans):
    a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    check(_lib.load().fpng_amd_1pass_layout(num_chans, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


HW_QUEUE_SOURCES = ("library_set", "caller_set", "driver_open", "hands_off")  # FPNG_AMD_HWQ_* (include/fpng_amd.h)


def runtime_info():
    """fpng_amd_runtime_info(): {"hw_queues", "hw_queue_source", "lanes"} -- the hardware queues the library believes the HIP runtime
    uses, why (it asks for eight when it is loaded before the process's first HIP call), and the lanes a new encoder would get."""
    info = _lib.RuntimeInfo()
    check(_lib.load().fpng_amd_runtime_info(C.byref(info)))
    return {"hw_queues": int(info.hw_queues), "hw_queue_source": HW_QUEUE_SOURCES[info.hw_queue_source], "lanes": int(info.lanes)}


def release_cached_memory():
    """Free the device buffers that destroyed encoders left with the library (fpng_amd_release_cached_memory)."""
    check(_lib.load().fpng_amd_release_cached_memory())


def pin_host_memory(arr):
    """Page-lock a numpy array's memory (fpng_amd_pin_host_memory): saves the runtime's pinning of every chunk it copies.
    Unpin before the array is freed."""
    check(_lib.load().fpng_amd_pin_host_memory(arr.ctypes.data, arr.nbytes))


def unpin_host_memory(arr):
    check(_lib.load().fpng_amd_unpin_host_memory(arr.ctypes.data))


def plan_bands(stats, w, h, num_chans, flags=0):
    """fpng_amd_plan_bands: stats = list of _lib.BandStats in row order -> (start_bits, BandPlan)."""
    n = len(stats)
    arr = (_lib.BandStats * n)(*stats)
    starts = (C.c_uint64 * n)()
    plan = _lib.BandPlan()
    check(_lib.load().fpng_amd_plan_bands(arr, n, w, h, num_chans, flags, starts, C.byref(plan)))
    return list(starts), plan


def band_window(is_first, is_last, start_bit, token_bits, eob_bits):
    """-> (file offset, bytes, shared head bytes) of the window fpng_amd_band_place() writes."""
    off, n, hd = C.c_uint64(0), C.c_size_t(0), C.c_uint32(0)
    check(_lib.load().fpng_amd_band_window(int(is_first), int(is_last), start_bit, token_bits, eob_bits, C.byref(off), C.byref(n), C.byref(hd)))
    return off.value, n.value, hd.value


def idat_crc_from_bands(raw, ends, zlib_size, adler):
    n = len(raw)
    return _lib.load().fpng_amd_idat_crc_from_bands((C.c_uint32 * n)(*raw), (C.c_uint64 * n)(*ends), n, zlib_size, adler)


def png_head(w, h, num_chans, zlib_size):
    b = (C.c_uint8 * 58)()
    check(_lib.load().fpng_amd_png_head(w, h, num_chans, zlib_size, b))
    return bytes(b)


def png_tail(adler, idat_crc):
    b = (C.c_uint8 * 20)()
    _lib.load().fpng_amd_png_tail(adler, idat_crc, b)
    return bytes(b)


class Node:
    """fpng_amd_node: one process, one encoder + staging ring per listed device, host batches dealt round-robin."""

    def __init__(self, devices):
        self.lib = _lib.load()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        check(self.lib.fpng_amd_node_create(C.byref(h), arr, len(devices)))
        self.h = h

    def size(self):
        return self.lib.fpng_amd_node_size(self.h)

    def encode_host_batch(self, images, flags=0, outs=None, paths=None, writer_threads=0):
        arr, sizes, keep = _host_batch_records(images, outs, paths)
        check(self.lib.fpng_amd_node_encode_host_batch(self.h, arr, len(images), flags, writer_threads))
        return [int(s) for s in sizes]

    def encode_host_image(self, image, w, h, num_chans, flags=0, out=None):
        """fpng_amd_node_encode_host_image: ONE host image (uint8 array) cut into row bands over the node's devices.  out=None ->
        the PNG as bytes (byte-identical to the single-device encoders' and to the reference's); out = a uint8 numpy array of at
        least max_encoded_size() bytes -> the file is written into it and its size returned (no allocation: a capture loop's form)."""
        b = _as_u8(image)
        if b.size < w * h * num_chans:
            raise ValueError("image buffer smaller than w*h*num_chans")
        buf = out if out is not None else np.empty(max_encoded_size(w, h, num_chans), dtype=np.uint8)
        assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"]

        def reserve(_user, nbytes):  # (one buffer that only ever "grows" inside its capacity: what was written stays)
            return buf.ctypes.data if nbytes <= buf.size else None

        cb = _lib.RESERVE_FN(reserve)
        n = C.c_size_t(0)
        check(self.lib.fpng_amd_node_encode_host_image(self.h, b.ctypes.data, w, h, num_chans, flags, cb, None, C.byref(n)))
        return n.value if out is not None else buf[: n.value].tobytes()

    def close(self):
        if getattr(self, "h", None):
            self.lib.fpng_amd_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _host_batch_records(images, outs, paths):
    n = len(images)
    arr = (HostImage * n)()
    sizes = (C.c_size_t * n)()
    keep = []
    for i, im in enumerate(images):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        keep.append(im)
        h, w, c = im.shape
        arr[i].pixels = im.ctypes.data
        arr[i].w, arr[i].h, arr[i].num_chans = w, h, c
        if outs is not None:
            arr[i].out = outs[i].ctypes.data
            arr[i].out_cap = outs[i].size
        arr[i].out_size = C.cast(C.byref(sizes, i * C.sizeof(C.c_size_t)), C.POINTER(C.c_size_t))
        if paths is not None:
            p = paths[i].encode() if isinstance(paths[i], str) else paths[i]
            keep.append(p)
            arr[i].path = p
    return arr, sizes, keep


def synth_image(kind, w, h, num_chans, seed=12345):
    """Deterministic test image as a uint8 array of shape (h, w, num_chans)."""
    out = np.empty(w * h * num_chans, dtype=np.uint8)
    check(_lib.load().fpng_amd_synth_image(SYNTH_KINDS[kind], seed, w, h, num_chans, out.ctypes.data))
    return out.reshape(h, w, num_chans)


class DecodeBatch:
    """What Encoder.make_decode_batch() returns: the files, the output tensors and the C arrays of one fpng_amd_decode_batch_device() call."""

    def __init__(self, pngs, outs, arr, res, desired_chans):
        self.pngs, self.outs, self.arr, self.res, self.desired_chans = pngs, outs, arr, res, desired_chans

    def statuses(self):
        """status of every file after the last call (one numpy view of the result records, no loop over them)"""
        return np.frombuffer(self.res, dtype=np.int32).reshape(-1, 4)[:, 3]

    def results(self):
        """list of (status, uint8 CUDA tensor (h, w, desired_chans) or None, channels_in_file)"""
        out, d = [], self.desired_chans
        for r, t in zip(self.res, self.outs):
            out.append((r.status, t.view(-1)[: r.w * r.h * d].view(r.h, r.w, d) if r.status == 0 else None, r.channels_in_file))
        return out


class Encoder:
    """Reusable device scratch + the HIP stream submissions are ordered against (one Encoder per thread).
    `stream="torch"` (default): every submit() is ordered behind the work already enqueued on torch's CURRENT
    stream of `device` at that moment (also when that is the default/null stream), so pixels produced by torch
    kernels need no host synchronisation; join() makes that stream wait for the outputs.  `stream="own"`: a
    private stream, the caller synchronises.  Anything else: a hipStream_t handle."""

    def __init__(self, device=0, stream="torch"):
        self.lib = _lib.load()
        self.device = device
        self._follow_torch = stream == "torch"
        h = C.c_void_p()
        if stream == "torch":
            with torch.cuda.device(device):
                sptr = torch.cuda.current_stream().cuda_stream
            check(self.lib.fpng_amd_encoder_create_on_stream(C.byref(h), device, C.c_void_p(sptr) if sptr else None))
        elif stream is None or stream == "own":
            check(self.lib.fpng_amd_encoder_create(C.byref(h), device, None))
        else:
            check(self.lib.fpng_amd_encoder_create_on_stream(C.byref(h), device, C.c_void_p(int(stream))))
        self.h = h
        self._keep = {}  # ticket -> buffers of that submission (at most 8 are in flight: the C side's slot ring)

    @property
    def lanes(self):
        """Lanes (stream + scratch set each) this encoder's submissions take turns over: fixed when it was created."""
        return int(self.lib.fpng_amd_encoder_lanes(self.h))

    def _sync_stream(self):
        if self._follow_torch:
            with torch.cuda.device(self.device):
                sptr = torch.cuda.current_stream().cuda_stream
            check(self.lib.fpng_amd_encoder_set_stream(self.h, C.c_void_p(sptr) if sptr else None))

    def close(self):
        if getattr(self, "h", None):
            self.lib.fpng_amd_encoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device-resident batch: the hot path ----
    @staticmethod
    def make_batch(images, outs):
        """Descriptor array (fpng_amd_image[n]) for a list of image / output tensors.  Build it once when the
        same buffers are encoded repeatedly (a capture pipeline): submit() then costs one C call."""
        n = len(images)
        arr = (Image * n)()
        for i, (im, out) in enumerate(zip(images, outs)):
            assert im.is_cuda and im.dtype == torch.uint8 and im.is_contiguous() and im.dim() == 3
            h, w, c = im.shape
            arr[i].d_pixels = im.data_ptr()
            arr[i].w, arr[i].h, arr[i].num_chans = w, h, c
            arr[i].d_out = out.data_ptr()
            arr[i].out_cap = out.numel()
        return (images, outs, arr)

    def submit(self, images, outs=None, flags=0):
        """images: list of uint8 CUDA tensors shaped (h, w, c), contiguous, and outs: list of uint8 CUDA
        tensors with >= max_encoded_size bytes -- or images = a make_batch() descriptor and outs = None.
        Asynchronous; call finish() for the sizes."""
        batch = images if outs is None else self.make_batch(images, outs)
        n = len(batch[2])
        self._sync_stream()
        t = C.c_uint64(0)
        check(self.lib.fpng_amd_encode_submit(self.h, batch[2], n, flags, C.byref(t)))
        self.last_ticket = t.value
        # buffers of submissions in flight stay alive; a submission 8 tickets back has finished (its slot was reused)
        self._keep[t.value] = batch
        for old in [k for k in self._keep if k + 8 <= t.value]:
            del self._keep[old]
        return n

    def wait(self, ticket, n):
        """Waits for the submission `ticket` (see last_ticket) only; returns its (png_size, mode, status) records."""
        res = (Result * n)()
        check(self.lib.fpng_amd_encode_wait(self.h, ticket, res, n))
        self._keep.pop(ticket, None)
        return [(r.png_size, r.mode, r.status) for r in res]

    def query(self, ticket):
        rc = self.lib.fpng_amd_encode_query(self.h, ticket)
        if rc < 0:
            check(rc)
        return bool(rc)

    def phase_names(self):
        """Names of the kernels/phases of the last submission's pipeline (see last_phase_ms)."""
        return self.lib.fpng_amd_encoder_phase_names(self.h).decode().split(",")

    def join(self):
        """Device-side join: the encoder's stream (torch's current one for stream="torch") waits for every
        submission made so far (no host wait)."""
        self._sync_stream()
        check(self.lib.fpng_amd_encoder_join(self.h))

    def finish(self, n):
        """Waits for ALL submissions in flight; returns (png_size, mode, status) of the last one's n images."""
        res = (Result * n)()
        check(self.lib.fpng_amd_encode_finish(self.h, res, n))
        self._keep = {}
        return [(r.png_size, r.mode, r.status) for r in res]

    def encode_tensors(self, images, flags=0, submissions=1):
        """Convenience: allocate outputs, encode, return list of PNG byte strings.  submissions: the batch goes to the GPU in that
        many submissions (at most one per image).  With frames in hand and nothing in flight, four submissions beat one -- their
        chains run on different lanes, one's assemble next to the next one's row walk: 8 x 8K 0.66 -> 0.56 ms, 64 x 1080p RGB
        0.34 -> 0.31 ms (tools/oneshot_split.py, profiles/r05_hw_queues.txt section 6); a caller that keeps submissions in flight
        anyway gains nothing from smaller ones."""
        outs = [torch.empty(max_encoded_size(im.shape[1], im.shape[0], im.shape[2]) + 64, dtype=torch.uint8,
                            device=im.device) for im in images]
        n = len(images)
        subs = min(max(1, submissions), 8)  # (the C side keeps the records of the last eight submissions)
        k = max(1, (n + subs - 1) // subs)
        parts = []
        for i in range(0, n, k):
            self.submit(images[i:i + k], outs[i:i + k], flags)
            parts.append((self.last_ticket, len(images[i:i + k])))
        res = []
        for ticket, m in parts:
            res += self.wait(ticket, m)
        pngs = []
        for out, (size, mode, status) in zip(outs, res):
            if status:
                raise FpngAmdError(status, "device reported an encode failure")
            pngs.append(bytes(out[:size].cpu().numpy()))
        return pngs, [m for _, m, _ in res]

    def decode_batch(self, pngs, desired_chans, dims=None):
        """fpng_amd_decode_batch: list of fpng-written files (bytes) -> list of (status, uint8 CUDA tensor (h, w, desired_chans)
        or None, channels_in_file).  Output buffers are sized from the files' headers (dims: optional list of (w, h) to skip that)."""
        import struct
        n = len(pngs)
        arr = (_lib.PngIn * n)()
        res = (_lib.DecodeResult * n)()
        keep, outs = [], []
        for i, p in enumerate(pngs):
            b = np.frombuffer(bytes(p), dtype=np.uint8)
            keep.append(b)
            w, h = dims[i] if dims else ((struct.unpack(">II", bytes(p[16:24]))) if len(p) >= 24 else (0, 0))
            cap = w * h * desired_chans if 0 < w <= (1 << 24) and 0 < h <= (1 << 24) and w * h <= (1 << 30) else 0
            t = torch.empty(max(cap, 16), dtype=torch.uint8, device=f"cuda:{self.device}")
            outs.append(t)
            arr[i].data = b.ctypes.data if b.size else None
            arr[i].size = b.size
            arr[i].d_pixels = t.data_ptr()
            arr[i].pixels_cap = t.numel()
        self._sync_stream()
        check(self.lib.fpng_amd_decode_batch(self.h, arr, n, desired_chans, res))
        out = []
        for i in range(n):
            r = res[i]
            ok = r.status == 0
            out.append((r.status, outs[i][: r.w * r.h * desired_chans].view(r.h, r.w, desired_chans) if ok else None, r.channels_in_file))
        return out

    def make_decode_batch(self, pngs, desired_chans, dims, outs=None):
        """Descriptor (fpng_amd_png_in[n] + the result records) for decode_device().  Build it once when the same device buffers are
        decoded repeatedly: a call then costs one C call, as make_batch() does for submit() -- filling the array takes Python about
        3 us a file, as long as the GPU needs for a 512 x 512 one."""
        n = len(pngs)
        arr = (_lib.PngIn * n)()
        res = (_lib.DecodeResult * n)()
        made = []
        for i, p in enumerate(pngs):
            w, h = dims[i]
            t = outs[i] if outs is not None else torch.empty(max(w * h * desired_chans, 16), dtype=torch.uint8, device=f"cuda:{self.device}")
            made.append(t)
            arr[i].data = p.data_ptr() if p.numel() else None
            arr[i].size = p.numel()
            arr[i].d_pixels = t.data_ptr()
            arr[i].pixels_cap = t.numel()
        return DecodeBatch(list(pngs), made, arr, res, desired_chans)

    def decode_device(self, pngs, desired_chans=None, dims=None, outs=None, results=True):
        """fpng_amd_decode_batch_device: list of uint8 CUDA tensors holding whole fpng-written files (e.g. the encoder's outputs) ->
        list of (status, uint8 CUDA tensor (h, w, desired_chans) or None, channels_in_file).  dims: list of (w, h) (sizes the
        output tensors; the files' bytes stay on the device); outs: optional preallocated uint8 CUDA tensors to decode into.
        pngs may be a make_decode_batch() descriptor (the other arguments are then its own); results=False: the call returns the
        descriptor, whose statuses() / results() can be asked later."""
        batch = pngs if isinstance(pngs, DecodeBatch) else self.make_decode_batch(pngs, desired_chans, dims, outs)
        self._sync_stream()
        check(self.lib.fpng_amd_decode_batch_device(self.h, batch.arr, len(batch.arr), batch.desired_chans, batch.res))
        return batch.results() if results else batch

    def last_decode_phase_ms(self):
        """{"sync", "offsets", "emit", "unfilter"} -> ms of the last decode call's kernels (first group of files), measured with HIP
        events on the encoder's stream while set_profiling(True)."""
        ms = (C.c_float * 4)()
        check(self.lib.fpng_amd_decode_last_phase_ms(self.h, C.byref(ms)))
        return dict(zip(("sync", "offsets", "emit", "unfilter"), (float(v) for v in ms)))

    def decode_host(self, png, desired_chans):
        """fpng_amd_decode_host: ONE fpng-written file (bytes) -> (status, uint8 numpy array (h, w, desired_chans) or None, channels_in_file);
        container checks, upload, GPU decode and one download into host memory (what fpng::fpng_decode_memory does for large images).
        status 64 (FPNG_AMD_DECODE_UNDECIDED): decode that file on the CPU."""
        b = np.frombuffer(bytes(png), dtype=np.uint8)
        res = _lib.DecodeResult()
        hold = []

        def reserve(_user, nbytes):
            hold[:] = [np.empty(nbytes, dtype=np.uint8)]
            return hold[0].ctypes.data

        cb = _lib.RESERVE_FN(reserve)
        self._sync_stream()
        check(self.lib.fpng_amd_decode_host(self.h, b.ctypes.data if b.size else None, b.size, desired_chans, cb, None, C.byref(res)))
        if res.status or not hold:
            return res.status, None, res.channels_in_file
        return 0, hold[0].reshape(res.h, res.w, desired_chans), res.channels_in_file

    def train_tables(self, images):
        """fpng_amd_train_tables: a new 1-pass table from a corpus of uint8 CUDA tensors (h, w, c), all with the same c, in the
        form the reference's training mode prints it (block prefix bytes as hex, pending bits, codes, code sizes)."""
        n = len(images)
        c = images[0].shape[2]
        arr = (Image * n)()
        for i, im in enumerate(images):
            assert im.is_cuda and im.dtype == torch.uint8 and im.is_contiguous() and im.dim() == 3
            arr[i].d_pixels = im.data_ptr()
            arr[i].w, arr[i].h, arr[i].num_chans = im.shape[1], im.shape[0], im.shape[2]
        prefix = (C.c_uint8 * 512)()
        nb, bb, bbs = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
        codes = (C.c_uint32 * 288)()
        sizes = (C.c_uint8 * 288)()
        self._sync_stream()
        check(self.lib.fpng_amd_train_tables(self.h, arr, n, c, prefix, 512, C.byref(nb), C.byref(bb), C.byref(bbs), codes, sizes))
        return {"prefix": bytes(prefix[: nb.value]).hex(), "bit_buf": bb.value, "bit_buf_size": bbs.value, "codes": list(codes),
                "code_sizes": list(sizes)}

    # ---- host buffers: what fpng_encode_image_to_memory() does ----
    def encode_host(self, image, w, h, num_chans, flags=0):
        b = _as_u8(image)
        if b.size < w * h * num_chans:
            raise ValueError("image buffer smaller than w*h*num_chans")
        cap = max_encoded_size(w, h, num_chans) if (w and h and num_chans in (3, 4)) else 64
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._sync_stream()
        check(self.lib.fpng_amd_encode_host(self.h, b.ctypes.data, w, h, num_chans, flags, out.ctypes.data, cap,
                                            C.byref(n)))
        return out[: n.value].tobytes()

    def encode_host_into(self, image, w, h, num_chans, out, flags=0):
        """Like encode_host() but into a caller-owned uint8 numpy array (>= max_encoded_size bytes), returning the
        PNG size: no allocation and no copy on the Python side, so a capture loop can reuse its buffers
        (3.6 ms instead of 18 ms per 8K frame: the one-shot form pays for a fresh 133 MB array and a bytes copy)."""
        b = _as_u8(image)
        if b.size < w * h * num_chans:
            raise ValueError("image buffer smaller than w*h*num_chans")
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        n = C.c_size_t(0)
        self._sync_stream()
        check(self.lib.fpng_amd_encode_host(self.h, b.ctypes.data, w, h, num_chans, flags, out.ctypes.data, out.size,
                                            C.byref(n)))
        return n.value

    def last_host_bands(self):
        """Row bands the last encode_host*() call was streamed in (1 = upload, encode, download one after the other)."""
        return self.lib.fpng_amd_encoder_last_host_bands(self.h)

    def encode_host_batch(self, images, flags=0, outs=None, paths=None, writer_threads=0):
        """Many host frames (uint8 arrays shaped (h, w, c)): uploads, encodes, downloads and file writes of consecutive
        frames overlap (fpng_amd_encode_host_batch).  outs: caller-owned uint8 arrays (>= max_encoded_size) or None when
        every frame goes to a file; paths: file names or None.  Returns the PNG sizes."""
        arr, sizes, keep = _host_batch_records(images, outs, paths)
        check(self.lib.fpng_amd_encode_host_batch(self.h, arr, len(images), flags, writer_threads))
        return [int(s) for s in sizes]

    def encode_host_growing(self, image, w, h, num_chans, flags=0):
        """fpng_amd_encode_host_to() with a growing bytearray as the output allocator (what the fpng:: drop-in does with
        its std::vector): returns (png bytes, list of the sizes the encoder asked for)."""
        b = _as_u8(image)
        buf = bytearray()
        asked = []
        hold = []

        def reserve(_user, nbytes):
            asked.append(nbytes)
            hold[:] = []  # (a bytearray cannot grow while a ctypes view of it is alive)
            if len(buf) < nbytes:
                buf.extend(bytes(nbytes - len(buf)))
            hold[:] = [(C.c_uint8 * len(buf)).from_buffer(buf)]
            return C.addressof(hold[0])

        cb = _lib.RESERVE_FN(reserve)
        n = C.c_size_t(0)
        self._sync_stream()
        rc = self.lib.fpng_amd_encode_host_to(self.h, b.ctypes.data, w, h, num_chans, flags, cb, None, C.byref(n))
        hold[:] = []
        check(rc)
        return bytes(buf[: n.value]), asked

    # ---- row bands (multi-GPU, one image): see include/fpng_amd.h ----
    @staticmethod
    def _band(rows, row_above, w, num_chans, y0, y1, h_total):
        for t in (rows, row_above):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous(), "band rows must be contiguous uint8 CUDA tensors"
        b = Band()
        b.d_rows = rows.data_ptr()
        b.d_row_above = row_above.data_ptr() if row_above is not None else None
        b.w, b.num_chans, b.y0, b.y1, b.h_total = w, num_chans, y0, y1, h_total
        return b

    def band_hist(self, band, d_hist):
        self._sync_stream()
        check(self.lib.fpng_amd_band_hist(self.h, C.byref(band), d_hist.data_ptr()))

    def band_encode(self, band, flags=0, d_hist=None):
        self._sync_stream()
        st = BandStats()
        check(self.lib.fpng_amd_band_encode(self.h, C.byref(band), flags, d_hist.data_ptr() if d_hist is not None else None, C.byref(st)))
        return st

    def band_place(self, band, start_bit, zlib_size, window):
        off, n = C.c_uint64(0), C.c_size_t(0)
        self._sync_stream()
        check(self.lib.fpng_amd_band_place(self.h, C.byref(band), start_bit, zlib_size, window.data_ptr(), window.numel(),
                                           C.byref(off), C.byref(n)))
        return off.value, n.value

    def band_crc_partials(self, device):
        """CRC partials (one uint32 per 64 KiB range of the file, as an int32 tensor) of the band placed last: XOR the
        ranks' arrays and hand the result to wrap_png()."""
        n = C.c_uint32(0)
        self._sync_stream()
        check(self.lib.fpng_amd_band_crc_partials(self.h, None, 0, C.byref(n)))  # (asks for the count)
        t = torch.empty(n.value, dtype=torch.int32, device=device)
        check(self.lib.fpng_amd_band_crc_partials(self.h, t.data_ptr(), n.value, C.byref(n)))
        return t

    def wrap_png(self, png_buf, zlib_size, adler, w, h, num_chans, crc_partials=None):
        n = C.c_size_t(0)
        self._sync_stream()
        if crc_partials is None:
            check(self.lib.fpng_amd_wrap_png(self.h, png_buf.data_ptr(), zlib_size, adler, w, h, num_chans, C.byref(n)))
        else:
            crc_partials = crc_partials.contiguous()
            self._keep_partials = crc_partials  # (asynchronous: keep it alive until the next call)
            check(self.lib.fpng_amd_wrap_png_crc(self.h, png_buf.data_ptr(), zlib_size, adler, w, h, num_chans,
                                                 crc_partials.data_ptr(), crc_partials.numel(), C.byref(n)))
        return n.value

    # ---- instrumentation ----
    def set_profiling(self, on=True):
        check(self.lib.fpng_amd_encoder_set_profiling(self.h, int(on)))

    def last_phase_ms(self):
        arr = (C.c_float * _lib.NUM_PHASES)()
        check(self.lib.fpng_amd_encoder_last_phase_ms(self.h, C.byref(arr)))
        return list(arr)


_default_encoder = None


def _encoder():
    global _default_encoder
    if _default_encoder is None:
        _default_encoder = Encoder(device=torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _default_encoder


def fpng_encode_image_to_memory(image, w, h, num_chans, flags=0):
    """Reference src/fpng.h:48.  Returns (ok, png_bytes); ok is False exactly where the reference
    returns false (bad dimensions / channel count, src/fpng.cpp:1670-1680).  Everything else
    (missing GPU, HIP errors) raises: it is not silently papered over."""
    try:
        return True, _encoder().encode_host(image, w, h, num_chans, flags)
    except FpngAmdError as e:
        if e.code == -1:  # FPNG_AMD_ERR_INVALID_ARG
            return False, b""
        raise


def fpng_encode_image_to_file(filename, image, w, h, num_chans, flags=0):
    """Reference src/fpng.h:52 / src/fpng.cpp:1806-1828."""
    ok, png = fpng_encode_image_to_memory(image, w, h, num_chans, flags)
    if not ok:
        return False
    try:
        with open(filename, "wb") as f:
            f.write(png)
    except OSError:
        return False
    return True


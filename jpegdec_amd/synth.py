"""Deterministic synthetic baseline-JPEG inputs (SURVEY.md 8d recipe).

Multi-octave value noise (cell size from max(W,H)/4 halving down to 4 px, amplitude x0.7 per
octave, bicubic upsample), normalised to 0..255, plus N(0, 3^2) pixel noise, encoded with
Pillow/libjpeg-turbo as baseline Huffman JPEG (quality 85, Annex-K tables unless optimize=True).
There is no network on the build or GPU boxes, so every test/bench input comes from here (or
from the small fixtures committed under tests/golden/).
"""
import io

import numpy as np


def _upsample(grid: np.ndarray, width: int, height: int, fast: bool) -> np.ndarray:
    if fast:
        import torch
        import torch.nn.functional as F

        t = torch.from_numpy(grid)[None, None]
        return F.interpolate(t, size=(height, width), mode="bicubic", align_corners=False)[0, 0].numpy()
    from PIL import Image

    return np.asarray(Image.fromarray(grid, mode="F").resize((width, height), Image.BICUBIC),
                      dtype=np.float32)


def value_noise_image(width: int, height: int, channels: int = 3, seed: int = 1234,
                      fast: bool = None) -> np.ndarray:
    """fast=True upsamples with torch (multi-threaded) instead of Pillow; default: only for
    images above 4 Mpixel (the two bicubic kernels differ slightly, so `fast` is part of the
    recipe: small test images always use Pillow)."""
    if fast is None:
        fast = width * height > (1 << 22)
    rng = np.random.default_rng(seed)
    acc = np.zeros((channels, height, width), dtype=np.float32)
    cell = max(width, height) // 4
    amp = 1.0
    while cell >= 4:
        gw = max(2, (width + cell - 1) // cell + 1)
        gh = max(2, (height + cell - 1) // cell + 1)
        for c in range(channels):
            grid = rng.random((gh, gw), dtype=np.float32)
            acc[c] += amp * _upsample(grid, width, height, fast)
        amp *= 0.7
        cell //= 2
    lo = acc.min()
    hi = acc.max()
    acc = (acc - lo) / max(hi - lo, 1e-6) * 255.0
    acc += rng.normal(0.0, 3.0, size=acc.shape).astype(np.float32)
    out = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    if channels == 1:
        return out[0]
    return np.ascontiguousarray(np.transpose(out, (1, 2, 0)))


def encode_jpeg(pixels: np.ndarray, quality: int = 85, subsampling="4:2:0", optimize: bool = False,
                restart_rows: int = 0, restart_blocks: int = 0, progressive: bool = False) -> bytes:
    """pixels: HxW (gray) or HxWx3 (RGB) uint8 -> baseline (or, progressive=True, libjpeg's default progressive
    script: first scan = the DC scan of all components, Al = 1) JPEG bytes."""
    from PIL import Image

    im = Image.fromarray(pixels)
    kw = dict(format="JPEG", quality=quality, optimize=optimize, progressive=progressive)
    if pixels.ndim == 3:
        kw["subsampling"] = {"4:4:4": 0, "4:2:2": 1, "4:2:0": 2}[subsampling]
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    if restart_blocks:
        kw["restart_marker_blocks"] = restart_blocks
    buf = io.BytesIO()
    im.save(buf, **kw)
    return buf.getvalue()


def widen_dqt(jpeg: bytes, factor: int) -> bytes:
    """Every DQT table of the file rewritten at word precision (Pq = 1, jpeg.inl:1742-1750) with its entries x factor
    (clipped to 65535): the reference takes such tables as they come, and with them max |coef| x max |q'| leaves the range the
    kernels' 24-bit multiplier covers -- the inputs that reach the 32-bit-multiply kernels."""
    out, i = bytearray(jpeg[:2]), 2
    while i < len(jpeg):
        assert jpeg[i] == 0xFF
        m = jpeg[i + 1]
        if m == 0xDA:
            out += jpeg[i:]
            break
        ln = (jpeg[i + 2] << 8) | jpeg[i + 3]
        seg = jpeg[i + 4:i + 2 + ln]
        if m == 0xDB:
            body, j = bytearray(), 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                n = 128 if pq else 64
                vals = [(seg[j + 1 + 2 * k] << 8) | seg[j + 2 + 2 * k] for k in range(64)] if pq else list(seg[j + 1:j + 65])
                body.append(0x10 | tq)
                for v in vals:
                    body += min(65535, v * factor).to_bytes(2, "big")
                j += 1 + n
            out += bytes([0xFF, 0xDB]) + (len(body) + 2).to_bytes(2, "big") + body
        else:
            out += jpeg[i:i + 2 + ln]
        i += 2 + ln
    return bytes(out)


def synth_jpeg(width: int, height: int, subsampling="4:2:0", seed: int = 1234, quality: int = 85,
               optimize: bool = False, restart_rows: int = 0, restart_blocks: int = 0, progressive: bool = False,
               dqt16: int = 0, noise: bool = False) -> bytes:
    """subsampling: '4:2:0' | '4:4:4' | '4:2:2' | '4:4:0' | 'gray'.  4:4:0 (luma sampled 1x2) goes through
    encode_jpeg_custom (restart_blocks = its restart interval in MCUs).
    noise: uniform pixel noise instead of the value-noise image (high contrast: long blocks, large coefficients).
    dqt16: the finished file's quantisers rewritten at word precision, x dqt16 (widen_dqt)."""
    ch = 1 if subsampling == "gray" else 3
    if noise:
        px = np.random.default_rng(seed).integers(0, 256, size=(height, width, ch) if ch == 3 else (height, width), dtype=np.uint8)
    else:
        px = value_noise_image(width, height, ch, seed)
    if subsampling == "4:4:0":
        jpeg = encode_jpeg_custom(px, quality, (1, 2), restart_interval=restart_blocks)
    else:
        jpeg = encode_jpeg(px, quality, subsampling if ch == 3 else None, optimize, restart_rows,
                           restart_blocks, progressive)
    return widen_dqt(jpeg, dqt16) if dqt16 else jpeg


def bits_per_pixel(jpeg: bytes, width: int, height: int) -> float:
    return 8.0 * len(jpeg) / float(width * height)


# ---- a small baseline encoder of our own -------------------------------------------------------
# Pillow/libjpeg-turbo cannot be asked for every sampling layout the decoder accepts (4:4:0 = luma
# sampled 1x2), and offers no control over the entropy coding.  This encoder (numpy DCT, Annex-K
# Huffman tables taken from a Pillow-written file, no optimisation) writes any luma H x V in {1,2} with
# 1x1 chroma, optional restart intervals, and is only meant for small test inputs.
def _annex_k_tables():
    """(quant_luma, quant_chroma at quality 50 in zigzag order, {(class,id): (bits[16], vals)}) parsed from a Pillow JPEG."""
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, format="JPEG", quality=50, optimize=False)
    d = buf.getvalue()
    q, huff = {}, {}
    i = 2
    while i < len(d):
        assert d[i] == 0xFF
        m = d[i + 1]
        if m == 0xDA:
            break
        ln = (d[i + 2] << 8) | d[i + 3]
        seg = d[i + 4:i + 2 + ln]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                q[seg[j] & 15] = np.frombuffer(seg[j + 1:j + 65], np.uint8).astype(np.int32)
                j += 65
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                bits = list(seg[j + 1:j + 17])
                n = sum(bits)
                huff[(tc, th)] = (bits, list(seg[j + 17:j + 17 + n]))
                j += 17 + n
        i += 2 + ln
    return q[0], q[1], huff


_ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
           28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
           47, 55, 62, 63]


def _codes(bits, vals):
    table, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            table[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return table


def encode_jpeg_custom(pixels: np.ndarray, quality: int = 85, luma_hv=(1, 2), restart_interval: int = 0,
                       dup_eob: bool = False, long_dc: bool = False, table_ids=((0, 0), (1, 1), (1, 1))) -> bytes:
    """Baseline JPEG of an HxWx3 RGB image with luma sampling factors luma_hv = (H, V) and 1x1 chroma:
    (1,1) 4:4:4, (2,1) 4:2:2, (1,2) 4:4:0, (2,2) 4:2:0.
    dup_eob: both AC tables code the end-of-block symbol twice (a second, 16-bit code) and every other block ends with
    the second one -- a malformed but decodable DHT (decoders with per-code LUTs do not notice).
    long_dc: both DC tables give categories 0-4 codes of 1-5 bits and categories 5-11 codes of ELEVEN bits, 11111000000 ..
    11111000110 -- legal, and codes that start 111110 and are longer than 10 bits are what the device pre-scan's 11-bit table
    key cannot tell apart (jda_dc_lut_walkable): such a file must stay on the serial pre-scan.
    table_ids: (DC table, AC table) of Y, Cb, Cr -- e.g. ((0, 0), (0, 1), (1, 1)): two components share a DC table and not their
    AC table (the walk's DC entries cannot know which AC symbol follows them: jda_wt_dc_follow)."""
    hs, vs = luma_hv
    h, w = pixels.shape[:2]
    rgb = pixels.astype(np.float64)
    ycc = [0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2],
           -0.168736 * rgb[..., 0] - 0.331264 * rgb[..., 1] + 0.5 * rgb[..., 2] + 128.0,
           0.5 * rgb[..., 0] - 0.418688 * rgb[..., 1] - 0.081312 * rgb[..., 2] + 128.0]
    mw, mh = 8 * hs, 8 * vs
    cx, cy = (w + mw - 1) // mw, (h + mh - 1) // mh
    planes = []
    for c, p in enumerate(ycc):
        p = np.pad(p, ((0, cy * mh - h), (0, cx * mw - w)), mode="edge")
        if c:                                   # box-filter the chroma down
            p = p.reshape(cy * mh // vs, vs, cx * mw // hs, hs).mean(axis=(1, 3))
        planes.append(p - 128.0)
    ql, qc, huff = _annex_k_tables()
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    qt = [np.clip((t * scale + 50) // 100, 1, 255) for t in (ql, qc)]
    k = np.arange(8)
    dct = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k[:, None] == 0, np.sqrt(1 / 8), 0.5)
    nat_q = []
    for t in qt:
        n = np.zeros(64, np.int32)
        n[_ZIGZAG] = t
        nat_q.append(n.reshape(8, 8))

    def blocks(p, q):
        hh, ww = p.shape
        b = p.reshape(hh // 8, 8, ww // 8, 8).transpose(0, 2, 1, 3)
        coef = np.einsum("ij,abjk,lk->abil", dct, b, dct)
        return np.rint(coef / q).astype(np.int32)

    cb = [blocks(planes[0], nat_q[0]), blocks(planes[1], nat_q[1]), blocks(planes[2], nat_q[1])]
    if long_dc:
        for th in (0, 1):
            huff[(0, th)] = ([1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 7, 0, 0, 0, 0, 0], list(range(12)))
    dc_t = [_codes(*huff[(0, 0)]), _codes(*huff[(0, 1)])]
    eob2 = [None, None]
    if dup_eob:
        for th in (0, 1):
            bits, vals = huff[(1, th)]
            bits = list(bits)
            bits[15] += 1                       # one more 16-bit code, the last of the table: symbol 0x00 again
            huff[(1, th)] = (bits, list(vals) + [0])
    ac_t = [_codes(*huff[(1, 0)]), _codes(*huff[(1, 1)])]
    if dup_eob:
        for th in (0, 1):
            eob2[th] = ac_t[th][0]              # (_codes keeps the last code of a repeated symbol)
            bits, vals = huff[(1, th)]
            ac_t[th][0] = _codes(bits[:15] + [bits[15] - 1], vals[:-1])[0]
    n_eob = [0]
    out = bytearray()
    acc, nacc = 0, 0

    def put(code, ln):
        nonlocal acc, nacc
        acc = (acc << ln) | (code & ((1 << ln) - 1))
        nacc += ln
        while nacc >= 8:
            byte = (acc >> (nacc - 8)) & 0xFF
            out.append(byte)
            if byte == 0xFF:
                out.append(0)
            nacc -= 8
        acc &= (1 << nacc) - 1

    def mag(v):
        a = abs(int(v))
        s = a.bit_length()
        return s, (int(v) if v >= 0 else int(v) + (1 << s) - 1)

    def emit_block(blk, c, pred):
        td, t = table_ids[c]
        zz = blk.reshape(64)[_ZIGZAG]
        s, bits = mag(zz[0] - pred)
        put(*dc_t[td][s])
        if s:
            put(bits, s)
        run = 0
        last = max([i for i in range(1, 64) if zz[i]] or [0])
        for i in range(1, last + 1):
            if zz[i] == 0:
                run += 1
                continue
            while run > 15:
                put(*ac_t[t][0xF0])
                run -= 16
            s, bits = mag(zz[i])
            put(*ac_t[t][(run << 4) | s])
            put(bits, s)
            run = 0
        if last < 63:
            n_eob[0] += 1
            put(*(eob2[t] if dup_eob and (n_eob[0] & 1) else ac_t[t][0]))
        return int(zz[0])

    pred = [0, 0, 0]
    n_mcu = 0
    rst = 0
    for my in range(cy):
        for mx in range(cx):
            if restart_interval and n_mcu and n_mcu % restart_interval == 0:
                if nacc:
                    put((1 << (8 - nacc)) - 1, 8 - nacc)
                out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                pred = [0, 0, 0]
            for by in range(vs):
                for bx in range(hs):
                    pred[0] = emit_block(cb[0][my * vs + by, mx * hs + bx], 0, pred[0])
            pred[1] = emit_block(cb[1][my, mx], 1, pred[1])
            pred[2] = emit_block(cb[2][my, mx], 2, pred[2])
            n_mcu += 1
    if nacc:
        put((1 << (8 - nacc)) - 1, 8 - nacc)

    def seg(marker, payload):
        return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)

    hdr = bytearray(b"\xff\xd8")
    hdr += seg(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    hdr += seg(0xDB, bytes([0]) + bytes(int(v) for v in qt[0]) + bytes([1]) + bytes(int(v) for v in qt[1]))
    hdr += seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3, 1, (hs << 4) | vs, 0, 2, 0x11, 1, 3, 0x11, 1]))
    for (tc, th), (bits, vals) in sorted(huff.items()):
        hdr += seg(0xC4, bytes([(tc << 4) | th]) + bytes(bits) + bytes(vals))
    if restart_interval:
        hdr += seg(0xDD, restart_interval.to_bytes(2, "big"))
    hdr += seg(0xDA, bytes([3, 1, (table_ids[0][0] << 4) | table_ids[0][1], 2, (table_ids[1][0] << 4) | table_ids[1][1],
                            3, (table_ids[2][0] << 4) | table_ids[2][1], 0, 63, 0]))
    # pad small files: the reference rejects anything under 256 bytes (jpeg.inl:1598)
    return bytes(hdr) + bytes(out) + b"\xff\xd9"

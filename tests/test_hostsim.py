"""Kernel logic without a GPU: tests/hostsim compiles the very code the HIP kernels run per lane
(jpegdec_amd/csrc/jda_device_core.h) with g++ and steps one wavefront lane by lane.  It must agree
byte for byte with the oracle, padded MCU area included.  (Test infrastructure, not a fallback.)"""
import ctypes as C
import os

import numpy as np
import pytest

import jpegdec_amd as J
from tests.cases import SYNTH_CASES, all_modes, jpeg_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["c444_600x16", "c444_333x217", "c420_1100x48", "gray_1600x16"])
def test_tile_order_does_not_matter(name, hostsim, oracle):
    """Tiles are decoded by independent wavefronts in any order: run them backwards and shrink the scan
    window so the HBM fall-back of the bit reader is exercised too."""
    jpeg = jpeg_for(name)
    hostsim.hostsim_set_reverse(1)
    hostsim.hostsim_set_window(64)
    try:
        for pt, opt in all_modes(name):
            rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
            assert hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
            assert np.array_equal(got, want), (name, pt, opt)
    finally:
        hostsim.hostsim_set_reverse(0)
        hostsim.hostsim_set_window(1 << 20)


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_wave_emulation_equals_oracle(name, hostsim, oracle):
    jpeg = jpeg_for(name)
    for pt, opt in all_modes(name):
        rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
        hrc = hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
        assert hrc == 0
        assert np.array_equal(got, want), (name, pt, opt, int(np.count_nonzero(got != want)))


@pytest.mark.parametrize("name", ["c420_640x368_rstrow", "gray_64x64_rst3", "c444_384x192_q100_rst7", "c420_512x256_q98_rstrow",
                                  "c422_1100x24_rstrow", "c440_300x64_rst5"])
def test_restart_interval_prescan_equals_oracle(name, hostsim, oracle):
    """a stream WITH restart intervals goes through the same segment walk as one without (one lane per 256 bytes, not one per
    interval): an interval's end is found by position (the filter recorded where every interval starts), the reference's rounding of
    ulBitOff without a refill and the predictor reset happen there, and the WRITE pass checks that the marker positions agree with the
    MCU count.  The index must equal the serial one entry for entry (reader phase, truncation flags, DC predictors, closing entry) and
    the decode must be the oracle's, byte for byte, in every pixel type and scale."""
    jpeg = jpeg_for(name)
    hostsim.hostsim_set_device_prescan(1)
    try:
        for pt, opt in all_modes(name):
            rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
            hrc = hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert hrc == 0 and hostsim.hostsim_prescan_used() == 2
            assert hostsim.hostsim_index_equal() == 1          # phase, DC predictor and truncation count of every block
            assert np.array_equal(got, want), (name, pt, opt)
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "gray_333x217", "c420_256x256_q98", "c444_256x256_q100_opt",
                                  "c420_1100x48", "gray_1600x16", "c420_16x16", "c444_8x8_q30", "c420_1280x720", "c422_333x217",
                                  "c440_200x120", "c420_250x250_q10"])
def test_markerless_prescan_equals_serial(name, hostsim, oracle):
    """without restart markers the per-block index is made by the segment walk (jda_seg_walk: one lane per 256
    bytes of the scan, speculative rounds until the decoder states at the segment boundaries stop changing, then the
    count pass, the host's sums and the write pass -- emulated here lane by lane exactly as jda_upload_batch + jda_segscan
    run them).  The index must equal the serial pre-scan's entry for entry (reader phase, DC predictor, closing entry,
    truncation count, 24-bit-multiply verdict) and the decode must be the oracle's."""
    jpeg = jpeg_for(name)
    hostsim.hostsim_set_device_prescan(1)
    try:
        for pt, opt in ((2, 0), (0, 2), (3, 8)):
            rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
            hrc = hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert hrc == 0 and hostsim.hostsim_prescan_used() == 2
            assert hostsim.hostsim_index_equal() == 1
            assert hostsim.hostsim_segscan_rounds() <= 20, hostsim.hostsim_segscan_rounds()      # self-synchronisation, not a serial crawl
            assert np.array_equal(got, want), (name, pt, opt)
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("name", ["c420_333x217", "c444_256x256_q100_opt", "gray_333x217", "c420_1280x720", "c422_333x217", "c420_256x256_q98",
                                  "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c440_300x64_rst5", "gray_64x64_rst3", "w16_c420_333x217_x400"])
def test_states_first_prescan_order_equals_serial(name, hostsim, oracle):
    """The order jda_upload_batch takes for a batch too small to fill the GPU (one image at a time; jda_launch_prescan_passes_ex): the
    entry states are settled by SPEC walks alone (round 0, round 1 over every segment, the work-list rounds), then ONE recording round
    walks every segment from its settled state.  The same index as the serial pre-scan's (and so as the other order's), the oracle's
    pixels; and on corrupted copies the same verdict -- an index that is made must be the serial one."""
    jpeg = jpeg_for(name)
    hostsim.hostsim_set_states_first(1)
    try:
        hostsim.hostsim_set_device_prescan(2)
        for pt, opt in ((2, 0), (0, 2), (3, 8)):
            rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
            hrc = hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert hrc == 0 and hostsim.hostsim_prescan_used() == 2
            assert hostsim.hostsim_index_equal() == 1
            assert np.array_equal(got, want), (name, pt, opt)
        base = bytearray(jpeg)
        sos = bytes(base).index(b"\xff\xda")
        rng = np.random.default_rng(23)
        for it in range(40):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            if (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan()):
                continue                                   # consumed more bits than the scan holds: out of contract
            rc, want, err = oracle.decode_canvas(jb, 2, 0)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jb, 2, 0)
            hrc = hostsim.hostsim_decode(jb, len(jb), 2, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert (rc == 1) == (hrc == 0), (name, it, rc, err, hrc)
            if hostsim.hostsim_prescan_used():
                assert hostsim.hostsim_index_equal() == 1, (name, it)
            if rc == 1:
                assert np.array_equal(got, want), (name, it)
    finally:
        hostsim.hostsim_set_device_prescan(0)
        hostsim.hostsim_set_states_first(0)


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "c422_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c440_300x64_rst5"])
def test_corrupted_scans_decode_like_the_oracle(name, hostsim, oracle):
    """The reference's fuzz idea (MacOS/JPEGDEC_Test/main.cpp:262-300) turned into a parity test: random byte
    corruptions inside the entropy-coded data.  A corrupted scan usually still "decodes" -- to garbage that
    exercises every odd path (runs past 63, 11-15 bit magnitudes, predictor wrap, stray markers) -- and the
    kernel logic must produce the oracle's garbage bit for bit, and fail on the same streams.
    Out of contract: a stream that runs out of data before the last MCU; the reference then
    decodes whatever its 2 KiB file buffer still holds, here the image ends with JPEG_DECODE_ERROR."""
    base = bytearray(jpeg_for(name))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(11)
    agree = 0
    for it in range(120):
        # two of three streams go through the device pre-scans (1: segment walk, restart streams one lane per interval; 2: restart
        # streams through the segment walk too, as the pipeline has it): a stream they cannot reproduce exactly must send them
        # back to the serial pre-scan, and an index they do make must be the serial one
        hostsim.hostsim_set_device_prescan(it % 3)
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
        jb = bytes(b)
        try:
            p = J.PreparedImage(jb)
        except J.JdaError:
            continue
        idx, nok = p.block_index()
        if (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan()):
            continue                                   # consumed more bits than the scan holds: out of contract
        for pt, opt in ((2, 0), (0, 2)):
            rc, want, err = oracle.decode_canvas(jb, pt, opt)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jb, pt, opt)
            hrc = hostsim.hostsim_decode(jb, len(jb), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert (rc == 1) == (hrc == 0), (name, it, pt, opt, rc, err, hrc)
            if hostsim.hostsim_prescan_used():
                assert hostsim.hostsim_index_equal() == 1, (name, it)
            if rc == 1:
                assert np.array_equal(got, want), (name, it, pt, opt)
                agree += 1
    hostsim.hostsim_set_device_prescan(0)
    assert agree >= 20


@pytest.mark.parametrize("luma_hv", [(2, 2), (1, 1), (2, 1)])
def test_duplicate_eob_code_takes_the_general_reader(luma_hv, hostsim, oracle):
    """P1 recognises EOB by comparing stream bits with the table's one EOB code.  A DHT that codes symbol 0x00 twice
    (malformed, but the reference's per-code LUTs decode it, jpeg.inl:1066-1275) is flagged by the host
    (JDA_DESC_GENERAL_P1) and decoded through the general bit reader: still the oracle's bytes."""
    from jpegdec_amd.synth import encode_jpeg_custom, value_noise_image
    jpeg = encode_jpeg_custom(value_noise_image(200, 120, 3, 77), 85, luma_hv, dup_eob=True)
    plain = encode_jpeg_custom(value_noise_image(200, 120, 3, 77), 85, luma_hv)
    p, q = J.PreparedImage(jpeg), J.PreparedImage(plain)
    assert p.general_p1() and not q.general_p1()
    p.close(); q.close()
    for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, J.SCALE_HALF), (J.GRAY8, J.SCALE_QUARTER)):
        rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert np.array_equal(got, want), (luma_hv, pt, opt)


@pytest.mark.parametrize("restart_interval", [0, 3])
def test_long_dc_codes_stay_on_the_serial_prescan(restart_interval, hostsim, oracle):
    """DC codes 111110.. of 11 bits: the segment walk's 11-bit key cannot stand for them (jda_dc_lut_walkable), so the front end
    must not defer such a file to the device pre-scan -- with or without restart intervals (round 2 checked the marker-less
    branch only) -- and the serial path decodes it to the oracle's bytes."""
    from jpegdec_amd.synth import encode_jpeg_custom, value_noise_image
    img = value_noise_image(200, 120, 3, 91)
    jpeg = encode_jpeg_custom(img, 85, (2, 2), restart_interval=restart_interval, long_dc=True)
    plain = encode_jpeg_custom(img, 85, (2, 2), restart_interval=restart_interval)
    p, q = J.PreparedImage(jpeg, device_prescan=True), J.PreparedImage(plain, device_prescan=True)
    assert not p.prescan_pending and q.prescan_pending
    p.close(); q.close()
    rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
    assert rc == 1
    got = np.full_like(want, 0x33)
    inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
    hostsim.hostsim_set_device_prescan(2)
    try:
        assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert hostsim.hostsim_prescan_used() == 0          # the walk was not taken
    finally:
        hostsim.hostsim_set_device_prescan(0)
    assert np.array_equal(got, want)


def test_reference_fixtures_with_restart_intervals_through_the_segment_walk(hostsim):
    """tulips (DRI) and the other fixtures with restart intervals: the segment walk's index == the serial pre-scan's"""
    from tests.ref_fixtures import GOOD, ref_jpeg
    hostsim.hostsim_set_device_prescan(2)
    n = 0
    try:
        for name in GOOD:
            jpeg = ref_jpeg(name)
            if J.parse(jpeg)["restart_interval"] == 0 or name == "corrupt5":
                continue
            p = J.PreparedImage(jpeg)
            geo = p.geometry(2, 0)
            got = np.zeros((geo["canvas_h"], geo["canvas_w"] * geo["bpp"]), np.uint8)
            assert hostsim.hostsim_decode(jpeg, len(jpeg), 2, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], p.info.mcus_x * p.info.mcu_w, p.info.mcus_y * p.info.mcu_h) == 0
            assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1, name
            p.close()
            n += 1
    finally:
        hostsim.hostsim_set_device_prescan(0)
    assert n >= 1


def _with_dri(jpeg: bytes, f):
    i = jpeg.index(b"\xff\xdd\x00\x04")
    v = (jpeg[i + 4] << 8) | jpeg[i + 5]
    nv = f(v)
    return jpeg[: i + 4] + bytes([nv >> 8, nv & 255]) + jpeg[i + 6:]


@pytest.mark.parametrize("name,f", [("c420_640x368_rstrow", lambda v: v * 2), ("c420_640x368_rstrow", lambda v: v - 1),
                                    ("c420_640x368_rstrow", lambda v: v + 1),          # same NUMBER of intervals (23), other places
                                    ("c420_512x256_q98_rstrow", lambda v: v + 1),
                                    ("c444_384x192_q100_rst7", lambda v: v + 1), ("gray_64x64_rst3", lambda v: 2)])
def test_restart_markers_that_disagree_with_the_mcu_count(name, f, hostsim, oracle):
    """The reference restarts by MCU COUNT (jpeg.inl:5339) and never looks where the markers were; the segment walk ends intervals
    where the filter found markers.  A DRI that does not match the markers makes the two disagree: the WRITE pass notices and the
    image takes the serial pre-scan -- pixels stay the reference's (garbage, but the same garbage)."""
    jpeg = _with_dri(jpeg_for(name), f)
    hostsim.hostsim_set_device_prescan(2)
    try:
        pt = 0 if name.startswith("gray") else 2
        rc, want, err = oracle.decode_canvas(jpeg, pt, 0)
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, 0)
        hrc = hostsim.hostsim_decode(jpeg, len(jpeg), pt, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
        assert hostsim.hostsim_prescan_used() == 0            # rejected by the device path
        if rc == 1:
            assert hrc == 0 and np.array_equal(got, want)
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("window", [1 << 20, 64])
def test_record_mode_flags_the_truncated_reads_of_real_photographs(window, hostsim, oracle):
    """RECORD mode of the device pre-scan (round 3: no WRITE walk): the counting walk leaves a record per block start and a
    candidate for every magnitude read that SOME entry lag of its segment truncates; finalize writes canonical entries, the
    candidates of the true lag become flagged entries with the reference reader's true phase.  The reference's own photographs
    have such reads: the index must agree with the serial pre-scan's -- bit position and flag of every block, the whole
    entry of a flagged one, predictors, the count of truncated reads -- and the decode must be the oracle's, through the window
    reader and (window = 64 bytes) through the general reader, which meets canonical entries there."""
    from tests.ref_fixtures import GOOD, ref_jpeg
    hostsim.hostsim_set_device_prescan(2)
    hostsim.hostsim_set_window(window)
    seen = 0
    try:
        for name in GOOD:
            jpeg = ref_jpeg(name)
            if J.parse(jpeg)["restart_interval"] != 0:
                continue                                   # (restart streams keep the counting walk + WRITE walk for now)
            for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, J.SCALE_QUARTER)):
                rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
                assert rc == 1
                got = np.full_like(want, 0x33)
                inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
                assert hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
                assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1, name
                assert np.array_equal(got, want), (name, pt, opt)
            p = J.PreparedImage(jpeg)
            if p.truncation_events() > 0:
                assert hostsim.hostsim_prescan_candidates() >= p.truncation_events()
                seen += 1
            p.close()
    finally:
        hostsim.hostsim_set_device_prescan(0)
        hostsim.hostsim_set_window(1 << 20)
    assert seen >= 2                                           # (the test means something: photographs with truncated reads went through)


def test_filter_state_machine_on_sixteen_bytes_at_once(hostsim):
    """jda_filter_classify / jda_filter_run (the marker filter's kernels evaluate JPEGFilter's two-state machine, jpeg.inl:1431-1540, on a
    16-bit "is FF" mask with one add) against the machine run byte by byte: every incoming state and valid count, on random groups
    that are dense in FF / 00 / RSTn bytes, and on every run length of FFs at every position."""
    rng = np.random.default_rng(3)
    alphabet = np.array([0xFF, 0xFF, 0xFF, 0x00, 0x00, 0xD0, 0xD7, 0xD8, 0xCF, 0x12, 0x80, 0xFE, 0x01], np.uint8)
    groups = [bytes(alphabet[rng.integers(0, len(alphabet), 16)]) for _ in range(4000)]
    for start in range(16):
        for run in range(1, 17 - start):
            g = bytearray(rng.integers(0, 0xFE, 16, dtype=np.uint8).tobytes())
            g[start:start + run] = b"\xff" * run
            groups.append(bytes(g))
    for g in groups:
        for valid in (16, 15, 9, 1, 0):
            for cin in (0, 1):
                assert hostsim.hostsim_filter_bits_check(g, valid, cin) == 0, (g.hex(), valid, cin)


@pytest.mark.parametrize("luma_hv,restart", [((2, 2), 0), ((1, 1), 0), ((2, 2), 7)])
def test_record_mode_on_the_densest_streams(luma_hv, restart, hostsim, oracle):
    """A flat image is the densest stream there is -- every block its two shortest codes, 32 bits per 4:2:0 MCU with the Annex K tables:
    some 390 block starts in a 256-byte segment.  The segments' record slots (jda_record_cap, from the tables' shortest codes) must hold
    them, the index must be the serial one and the picture the oracle's."""
    from jpegdec_amd.synth import encode_jpeg_custom
    flat = np.full((256, 640, 3), 117, np.uint8)
    flat[100:140, 300:360] = 30                                   # (a little structure, so that not every segment is the same)
    jpeg = encode_jpeg_custom(flat, 90, luma_hv, restart_interval=restart)
    hostsim.hostsim_set_device_prescan(2)
    try:
        rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1
        assert np.array_equal(got, want)
        p = J.PreparedImage(jpeg)
        assert p.n_blocks * 256 // max(len(p.scan()), 1) > 200          # (the test means something: hundreds of blocks per segment)
        p.close()
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("name", ["c420_1280x720", "c444_333x217", "c420_256x256_q98", "gray_333x217", "c444_256x256_q100_opt"])
def test_two_symbols_a_step_changes_nothing_but_the_number_of_steps(name, hostsim, oracle):
    """The walk's tables hold, behind an AC symbol, the symbol that follows it where the ten key bits contain that one's code too
    (jda_wt_pair): a step then takes both.  Same index, same picture as the walk symbol by symbol -- in a third fewer steps."""
    jpeg = jpeg_for(name)
    hostsim.hostsim_walk_steps.restype = C.c_ulonglong
    hostsim.hostsim_set_device_prescan(2)
    try:
        rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
        steps = []
        for off in (1, 0):
            hostsim.hostsim_walk_steps(off)
            got = np.full_like(want, 0x33)
            assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
            assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1
            assert np.array_equal(got, want)
            steps.append(hostsim.hostsim_walk_steps(0))
        assert steps[1] < (0.97 if "q100" in name else 0.8) * steps[0], steps      # (q100: long codes, large magnitudes -- few pairs)
    finally:
        hostsim.hostsim_walk_steps(0)
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("table_ids", [((0, 0), (0, 1), (1, 1)), ((0, 1), (1, 0), (1, 0)), ((1, 1), (0, 0), (0, 1))])
def test_components_that_share_a_dc_table_and_not_their_ac_table(table_ids, hostsim, oracle):
    """A DC entry of the walk's tables takes the first AC symbol of its block along -- from the AC table of the components that use
    the DC table (jda_wt_dc_follow).  Where two of them use different AC tables there is no such table: those DC entries stay
    single; crossed assignments (DC 0 with AC 1) follow the cross.  Index and picture as ever."""
    from jpegdec_amd.synth import encode_jpeg_custom
    rng = np.random.default_rng(5)
    img = np.clip(rng.normal(128, 40, (96, 160, 3)) + np.linspace(0, 60, 160)[None, :, None], 0, 255).astype(np.uint8)
    jpeg = encode_jpeg_custom(img, 80, (2, 2), table_ids=table_ids)
    hostsim.hostsim_set_device_prescan(2)
    try:
        rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1
        assert np.array_equal(got, want)
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("luma_hv,restart,quality", [((2, 2), 3, 100), ((2, 2), 7, 100), ((1, 1), 5, 100), ((2, 1), 2, 98), ((2, 2), 1, 100)])
def test_truncated_reads_in_streams_with_restart_intervals(luma_hv, restart, quality, hostsim, oracle):
    """Noise at quality 98-100: magnitude reads the reference truncates, in intervals of a few MCUs -- the byte lags behind an
    interval's closing EOB (whose refill waits for the rounding) must come out right for the flags to: whole index entries compared."""
    from jpegdec_amd.synth import encode_jpeg_custom
    rng = np.random.default_rng(restart * 10 + quality)
    img = rng.integers(0, 256, (112, 176, 3)).astype(np.uint8)
    jpeg = encode_jpeg_custom(img, quality, luma_hv, restart_interval=restart)
    p = J.PreparedImage(jpeg)
    assert p.truncation_events() >= 2                             # (the test means something)
    p.close()
    hostsim.hostsim_set_device_prescan(2)
    try:
        rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert hostsim.hostsim_prescan_used() == 2 and hostsim.hostsim_index_equal() == 1
        assert np.array_equal(got, want)
    finally:
        hostsim.hostsim_set_device_prescan(0)


def test_pair_halves_of_the_walk_tables_against_two_single_lookups(hostsim):
    """jda_wt_pair: for every short key of the four walk tables that has a pair half, and random continuations of the stream behind the
    key's ten bits, the symbol a walk decodes behind the first -- by its own key derivation at that place, long codes included -- is the
    one the pair half describes.  Annex K tables, optimised tables (Pillow's), tables with an 11-bit DC code and crossed assignments."""
    from jpegdec_amd.synth import encode_jpeg_custom
    hostsim.hostsim_walk_pairs_check.restype = C.c_long
    rng = np.random.default_rng(11)
    img = np.clip(rng.normal(128, 50, (64, 96, 3)), 0, 255).astype(np.uint8)
    files = [jpeg_for("c420_1280x720"), jpeg_for("c444_256x256_q100_opt"), jpeg_for("gray_333x217"), jpeg_for("c420_250x250_q10"),
             encode_jpeg_custom(img, 85, (2, 2), table_ids=((0, 1), (1, 0), (1, 0))),
             encode_jpeg_custom(img, 85, (1, 1), table_ids=((0, 0), (0, 1), (1, 1))),
             encode_jpeg_custom(img, 90, (2, 2), dup_eob=True)]
    for i, f in enumerate(files):
        n = hostsim.hostsim_walk_pairs_check(f, len(f), 48, i)
        assert n > 20000, (i, n)                                  # (hundreds of paired keys per table, each under 48 continuations)


def test_a_damaged_interval_that_reaches_its_marker_behind_the_images_last_block(hostsim, oracle):
    """Found by tools/gpu_fuzz_pipeline.py (seed 40404): one changed byte in the interval before the last restart marker; the damaged
    interval decodes to more blocks than it had, so the reference -- which restarts by MCU count and never looks for markers -- rounds
    up and resets 2,000 bits before the place the marker stood at, while a walk that follows the markers reaches the marker only
    with the image's block count used up.  "Behind the image's last block, nobody counts" let that pass: the marker's count (3,444)
    is inside the image, so it must be met exactly (jda_rst_event_item) -- the image goes to the serial pre-scan."""
    b = bytearray(jpeg_for("c444_384x192_q100_rst7"))
    assert b[121832] == 0x65
    b[121832] = 0x2a
    jpeg = bytes(b)
    hostsim.hostsim_set_device_prescan(2)
    try:
        rc, want, err = oracle.decode_canvas(jpeg, J.RGB8888, 0)
        assert rc == 1
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, J.RGB8888, 0)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
        assert hostsim.hostsim_prescan_used() == 0               # (the walk's index is not the reference's: rejected)
        assert np.array_equal(got, want)
    finally:
        hostsim.hostsim_set_device_prescan(0)


@pytest.mark.parametrize("name", ["c444_384x192_q100_rst7", "c420_640x368_rstrow", "c440_300x64_rst5", "c420_512x256_q98_rstrow"])
def test_corrupted_restart_streams_through_the_record_mode_walk(name, hostsim, oracle):
    """More of the corruption test for the streams whose walk follows markers the reference never looks at: every stream through the
    segment walk in RECORD mode (what the pipeline runs); an index it accepts must be the serial one, the picture the oracle's."""
    base = bytearray(jpeg_for(name))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(40404)
    used = agree = 0
    hostsim.hostsim_set_device_prescan(2)
    try:
        for it in range(150):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                # (half of the changes in the stream's last tenth: where an interval's damage meets the end of the image)
                lo = sos + 14 if it % 2 else len(b) - max((len(b) - sos) // 10, 40)
                b[int(rng.integers(lo, len(b) - 2))] = int(rng.integers(0, 256))
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            if (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan()):
                continue
            rc, want, err = oracle.decode_canvas(jb, J.RGB8888, 0)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jb, J.RGB8888, 0)
            hrc = hostsim.hostsim_decode(jb, len(jb), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert (rc == 1) == (hrc == 0), (name, it, rc, err, hrc)
            if hostsim.hostsim_prescan_used():
                used += 1
                assert hostsim.hostsim_index_equal() == 1, (name, it)
            if rc == 1:
                assert np.array_equal(got, want), (name, it)
                agree += 1
    finally:
        hostsim.hostsim_set_device_prescan(0)
    assert agree >= 30 and used >= 10, (agree, used)


@pytest.mark.parametrize("name", ["c420_640x368_rstrow", "c444_384x192_q100_rst7", "c440_300x64_rst5", "gray_64x64_rst3"])
def test_damaged_restart_markers_through_the_record_mode_walk(name, hostsim, oracle):
    """The reference counts MCUs and never looks for restart markers (jpeg.inl:5337-5348); the segment walk follows them.  Markers
    deleted, doubled, renumbered, inserted in the middle of an interval, moved by a few bytes, another DRI value: a stream the walk's
    checks accept must come out with the serial pre-scan's index, every picture and status as the oracle's (tools/cpu_fuzz_markers.py
    is the long version of this test)."""
    base = bytearray(jpeg_for(name))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(5)

    def markers(b):
        out, i = [], sos + 2
        while i < len(b) - 1:
            if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7:
                out.append(i); i += 2
            else:
                i += 1
        return out

    used = agree = 0
    hostsim.hostsim_set_device_prescan(2)
    try:
        for it in range(90):
            b = bytearray(base)
            ms = markers(b)
            m = ms[int(rng.integers(0, len(ms)))]
            kind = it % 6
            if kind == 0:
                del b[m:m + 2]
            elif kind == 1:
                b[m:m] = b[m:m + 2]
            elif kind == 2:
                b[m + 1] = 0xD0 + int(rng.integers(0, 8))
            elif kind == 3:
                at = int(rng.integers(sos + 14, len(b) - 2))
                b[at:at] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
            elif kind == 4:
                mk = bytes(b[m:m + 2]); del b[m:m + 2]
                at = max(sos + 14, min(len(b) - 2, m + int(rng.integers(-6, 7))))
                b[at:at] = mk
            else:
                i = bytes(b).find(b"\xff\xdd\x00\x04")
                v = max(1, ((b[i + 4] << 8) | b[i + 5]) + int(rng.integers(-2, 3)))
                b[i + 4], b[i + 5] = v >> 8, v & 255
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            if (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan()):
                continue
            rc, want, err = oracle.decode_canvas(jb, J.RGB8888, 0)
            got = np.full_like(want, 0x33)
            inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jb, J.RGB8888, 0)
            hrc = hostsim.hostsim_decode(jb, len(jb), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
            assert (rc == 1) == (hrc == 0), (name, it, kind, rc, err, hrc)
            if hostsim.hostsim_prescan_used():
                used += 1
                assert hostsim.hostsim_index_equal() == 1, (name, it, kind)
            if rc == 1:
                assert np.array_equal(got, want), (name, it, kind)
                agree += 1
    finally:
        hostsim.hostsim_set_device_prescan(0)
    assert agree >= 20, (agree, used)


def test_p1_in_chunks_on_the_host_simulator(hostsim, oracle):
    """P1's chunked mode (jda_p1c_block / _item / _finish, jda_decode_chunk_win) stepped lane by lane over the serial pre-scan's
    continuation entries (one every 8 AC symbols: jda_image_block_cont): pass A for every block, then every entry of the tile as an item
    of its own -- blocks flagged for truncated reads decoded whole, restart streams, every layout -- bit-exact with the oracle, and the
    photographs really do go through chunks (thousands of items)."""
    from tests.ref_fixtures import ref_jpeg
    hostsim.hostsim_chunk_items.restype = C.c_ulonglong
    hostsim.hostsim_set_chunked(1)
    try:
        hostsim.hostsim_chunk_items()
        cases = [(n, jpeg_for(n)) for n in ("c420_333x217", "c444_256x256_q100_opt", "c422_333x217", "c440_200x120", "gray_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7")]
        cases += [("ref:" + n, ref_jpeg(n)) for n in ("tulips", "zebra", "perf")]
        items = 0
        for name, jpeg in cases:
            gray = oracle.info(jpeg)["ncomp"] == 1
            for pt, opt in ((2, 0), (1, 2)) if gray else ((3, 0), (1, 0), (3, 2)):
                rc, want, err = oracle.decode_canvas(jpeg, pt, opt)
                assert rc == 1
                got = np.full_like(want, 0x33)
                inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jpeg, pt, opt)
                assert hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh) == 0
                assert np.array_equal(got, want), (name, pt, opt, int(np.count_nonzero(got != want)))
            items += hostsim.hostsim_chunk_items()
        assert items > 20000
    finally:
        hostsim.hostsim_set_chunked(0)

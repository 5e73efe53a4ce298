#!/usr/bin/env python3
"""Corrupted streams through jda_upload_batch's device pre-scan (test infrastructure: the serial host pre-scan and the oracle are the
checkers): ONE image at a time -- the states-first order of the rounds -- and in batches of a few dozen (the other order):
the index the device hands back must be the serial pre-scan's (jda_index_equivalent, DC values, MCU count) whether the device made it or
handed the stream back, and the decode the oracle's.  Usage (GPU box): python tools/gpu_fuzz_upload.py [rounds] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jpegdec_amd as J  # noqa: E402
from oracle.loader import OracleDecoder  # noqa: E402
from tests.cases import jpeg_for  # noqa: E402
from tests.ref_fixtures import ref_jpeg  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 91)
ctx = J.Context(0)
oracle = OracleDecoder()
bases = ("c420_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c422_333x217", "c420_1280x720", "gray_333x217", "c420_256x256_q98",
         "ref:zebra", "ref:tulips", "ref:sciopero", "w16_c420_333x217_x400")
total = on_device = decoded = 0


def check(jpegs, dimgs, preps):
    global total, on_device, decoded
    for j, p_, d_ in zip(jpegs, preps, dimgs):
        host = J.PreparedImage(j, flags=J.PREPARE_SERIAL_PRESCAN)
        want_idx, nok = host.block_index()
        got_idx, got_dc = d_.read_index()
        nb = nok * p_.info.blocks_per_mcu
        assert d_.n_mcus_ok == nok, "MCU count"
        assert J.index_equivalent(got_idx[:nb], want_idx[:nb]) and np.array_equal(got_dc[:nb], host.block_dc()[:nb]), "index"
        total += 1; on_device += int(bool(d_.prescan_on_device))
        if nok == p_.n_mcus:                                     # a whole image: the oracle's pixels
            rc, want, _ = oracle.decode_canvas(j, J.RGB8888, 0)
            g = p_.geometry(J.RGB8888, 0)
            pitch = (want.shape[1] + 15) // 16 * 16
            out = ctx.malloc(pitch * want.shape[0])
            b = J.Batch(ctx, [d_], [(out, pitch, g["canvas_w"], g["canvas_h"])], [J.RGB8888], [0])
            b.decode(); ctx.sync()
            got = ctx.to_host(out, pitch * want.shape[0]).reshape(want.shape[0], pitch)[:, : want.shape[1]]
            assert rc == 1 and np.array_equal(got, want), "pixels"
            decoded += 1
            b.close(); ctx.free(out)
        host.close(); d_.close()


for r in range(rounds):
    batch = []
    for name in bases:
        base = bytearray(ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name))
        sos = bytes(base).index(b"\xff\xda")
        made = 0
        while made < 4:
            b = bytearray(base)
            for _ in range(int(rng.integers(0, 4))):             # (0: the file as it is)
                b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            out_of_contract = (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan())      # (ran out of data)
            p.close()
            if out_of_contract:
                continue
            batch.append(jb); made += 1
    for jb in batch[::2]:                                        # one image at a time: the states-first order
        p = J.PreparedImage(jb, device_prescan=True)
        check([jb], [J.DeviceImage(ctx, p)], [p])
    many = batch * 16                                            # > 131,072 segments in all (some 250,000): the other order
    preps = [J.PreparedImage(j, device_prescan=True) for j in many]
    check(many, J.upload_batch(ctx, preps), preps)
print("upload path: %d streams, %d indexed on the device, %d decoded whole -- every index the serial pre-scan's, every decode the oracle's" % (total, on_device, decoded))
ctx.close()

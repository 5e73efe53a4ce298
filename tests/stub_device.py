"""A stand-in for the DEVICE half of jpegdec_amd (context, resident images, launch plans, pipeline) so that bench.py's own
control flow -- sharding one image list, host placement, the collectives, the exactly-once proof, the JSON record -- can be
run under gloo on a box without a GPU (tests/test_bench_flow_gloo.py).  The host half (parse, tables, pre-scan) is the real
library.  Nothing is decoded here: a surface's "checksum" is a hash of the file it would have been decoded from."""
import hashlib
import time

import jpegdec_amd as _J

RGB565_LE, RGB565_BE, RGB8888, GRAY8 = _J.RGB565_LE, _J.RGB565_BE, _J.RGB8888, _J.GRAY8
PreparedImage = _J.PreparedImage
prepare_batch = _J.prepare_batch
surface_checksum_host = _J.surface_checksum_host


class _Lib:
    @staticmethod
    def jda_device_count():
        return 1


def load_library():
    return _Lib()


def _key(prepared):
    return hashlib.sha256(prepared.scan().tobytes() + bytes(prepared.tables())).digest()[:8]


class Context:
    def __init__(self, device=0):
        self.device = device
        self._next = 1 << 20
        self.surfaces = {}          # device "pointer" -> file key written there
        self._t = [0.0, 0.0]

    def pci_bus_id(self):
        return "0000:00:00.0"

    def malloc(self, n):
        p = self._next
        self._next += (n + 255) & ~255
        return p

    def free(self, p):
        pass

    def sync(self):
        pass

    def timer_start(self):
        self._t[0] = time.perf_counter()

    def timer_stop(self):
        self._t[1] = time.perf_counter()

    def timer_elapsed_ms(self):
        return max((self._t[1] - self._t[0]) * 1e3, 1e-3)

    def checksums(self, surfaces, row_bytes):
        return [int.from_bytes(self.surfaces[s[0]], "little") for s in surfaces]

    def to_host(self, ptr, n):
        raise RuntimeError("the stub holds no pixels")

    def close(self):
        pass


class _Dev:
    def __init__(self, prepared):
        self.key = _key(prepared)
        self.prescan_on_device = False

    def close(self):
        pass


def upload_batch(ctx, prepared_list):
    return [_Dev(p) for p in prepared_list]


class Batch:
    def __init__(self, ctx, images, outputs, pixel_types, options):
        self.ctx, self.images, self.outputs = ctx, images, outputs
        self.stats = {"source_pixels": 0, "output_bytes": sum(o[1] * o[3] for o in outputs), "scan_bytes": 0, "index_bytes": 0,
                      "table_bytes": 0, "n_launches": 1, "n_workgroups": 1}
        self.decodes = 0

    def decode(self):
        self.decodes += 1
        for im, o in zip(self.images, self.outputs):
            self.ctx.surfaces[o[0]] = im.key

    def close(self):
        pass


class Pipeline:
    def __init__(self, ctx, max_images, depth=2, host_threads=0):
        self.ctx, self.n, self.batches = ctx, {}, 0

    @staticmethod
    def pack(jpegs, outputs, pixel_types, options):
        return (jpegs, outputs, pixel_types, options)

    def submit_packed(self, packed):
        self.batches += 1
        self.n[self.batches] = len(packed[0])
        return self.batches

    def submit(self, jpegs, outputs, pixel_types, options):
        return self.submit_packed(self.pack(jpegs, outputs, pixel_types, options))

    def wait(self, ticket):
        return [0] * self.n.pop(ticket)

    @property
    def stats(self):
        return {"device_images": 0, "host_path_images": 0}

    def close(self):
        pass

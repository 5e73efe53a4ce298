import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import jpegdec_amd as J
j = open("tests/golden/ref/tulips.jpg", "rb").read()
for flags in (J.PREPARE_SERIAL_PRESCAN, 0, J.PREPARE_SERIAL_PRESCAN, 0):
    for _ in range(20): J.PreparedImage(j, flags=flags).close()
    t = time.perf_counter()
    for _ in range(300): J.PreparedImage(j, flags=flags).close()
    print("tulips prepare flags", flags, "%.1f us" % ((time.perf_counter() - t) / 300 * 1e6))

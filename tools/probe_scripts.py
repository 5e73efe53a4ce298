"""replay tests/golden/script_walks.json through the product class, one process per script (finds the one that kills the process)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from oracle.loader import RefDecoder
    from tests.cases import jpeg_for
    from tests.ref_fixtures import ref_jpeg
    G = json.load(open(os.environ.get("SCRIPTS", os.path.join(ROOT, "tests", "golden", "script_walks.json"))))
    item = G["scripts"][int(sys.argv[1])]
    sc = item["script"]
    jpeg = ref_jpeg(sc["image"][4:]) if sc["image"].startswith("ref:") else jpeg_for(sc["image"])
    p = RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))
    got = p.run_script(jpeg, sc["ops"])
    want = item["ref"].get("values")
    print("OK" if got == want else "DIFF", sc["i"], sc["image"], sc["ops"], "got", got, "want", want)
else:
    G = json.load(open(os.environ.get("SCRIPTS", os.path.join(ROOT, "tests", "golden", "script_walks.json"))))
    for k, item in enumerate(G["scripts"]):
        if "crashed" in item["ref"]:
            continue
        r = subprocess.run([sys.executable, __file__, str(k)], capture_output=True, text=True)
        if r.returncode != 0:
            print("CRASH", k, item["script"]["image"], item["script"]["ops"], r.returncode, r.stderr[-300:])
        elif not r.stdout.startswith("OK"):
            print(r.stdout.strip()[:1500])

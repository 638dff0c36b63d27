#!/usr/bin/env python3
"""random generator options x sizes x seeds: the device headers compiled for the CPU (build/libhostsim.so) against the
reference (oracle/_ref). CPU only; a cheap way to look for parity bugs between GPU runs.  python tools/fuzz_parity.py [n] [seed]
With a third argument "gpu" the product library decodes instead (public API, needs an MI355X)."""
import ctypes as C, os, random, signal, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from streams import SYNTH
from refdec import Ref


def synth(mode, w, h, seed, **opts):
    """like streams.synth, but without leaving the stream in the on-disk cache (hundreds of them would travel to the GPU box)"""
    with tempfile.NamedTemporaryFile(suffix=".jxl") as tmp:
        subprocess.run([SYNTH, mode, str(w), str(h), str(seed), tmp.name] + ["%s=%s" % kv for kv in sorted(opts.items())], check=True, stderr=subprocess.DEVNULL)
        return open(tmp.name, "rb").read()


def decode_in_child(ref, d, limit=60):
    """ref.decode(d) in a forked child: damaged streams can crash the reference itself (it segfaults on some flips in a permuted TOC's
    sections); that is reported as "CRSH", not as the end of the sweep."""
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            signal.alarm(limit)   # ... or hang it (a flipped jxlp box size does): the alarm ends the child
            err, px = ref.decode(d)
            with os.fdopen(wr, "wb") as f:
                f.write(("%-4s" % err).encode("latin1"))
                if err == "": f.write(np.array(px.shape[:2], np.int32).tobytes() + np.ascontiguousarray(px).tobytes())
        finally:
            os._exit(0)
    os.close(wr)
    with os.fdopen(rd, "rb") as f: data = f.read()
    _, status = os.waitpid(pid, 0)
    if status != 0 or len(data) < 4: return "CRSH", None
    err = data[:4].decode("latin1").strip()
    if err: return err, None
    hh, ww = np.frombuffer(data[4:12], np.int32)
    return "", np.frombuffer(data[12:], np.uint8).reshape(hh, ww, 4)


def pick_vardct(r):
    o = {}
    if r.random() < .3: o["bctx"] = 1
    if r.random() < .3: o["presets"] = r.choice([2, 3])
    if r.random() < .3: o["orders"] = 1
    if r.random() < .3: o["passes"] = r.choice([2, 3])
    if r.random() < .2: o["cfl"] = 1
    if r.random() < .2: o["simpleclusters"] = 1; o["logalpha"] = r.choice([5, 6, 7, 8])
    if r.random() < .2: o["hfprefix"] = 1
    if r.random() < .2: o["hflz77"] = 1
    if r.random() < .2: o["dq"] = r.choice([1, 2])
    if r.random() < .2: o["permute"] = 1
    if r.random() < .15: o["container"] = r.choice([1, 2])
    if r.random() < .3: o["maxlog"] = r.choice([4, 5, 6, 7, 8])
    if r.random() < .15 and "passes" not in o:
        o["alpha"] = 1
        if r.random() < .4: o["fullheader"] = 1; o["noxyb"] = r.choice([0, 1])
    elif r.random() < .15: o["bpp"] = r.choice([9, 10, 12, 15])
    # an encode of a picture instead of coefficient-domain synthesis (the generator takes it with one pass, transforms up to 64x64,
    # default chroma-from-luma)
    if r.random() < .3 and "passes" not in o and "cfl" not in o and o.get("maxlog", 6) <= 6:
        o["forward"] = 1
        if r.random() < .5: o["detail"] = r.choice([0.5, 1, 3, 5]); o["beta"] = r.choice([0.1, 0.35, 0.8])
    return o


def pick_modular(r):
    o = {}
    if r.random() < .6: o["tree"] = r.choice([1, 2, 3, 5, 5])   # (5: 48 leaves over every property and predictor but the weighted one)
    p = r.random()
    if p < .25: o["palette"] = r.choice([1, 2, 3])
    elif p < .5: o["rct"] = r.choice([-1] + list(range(42)))
    if r.random() < .3: o["prefix"] = 1
    if r.random() < .3: o["lz77"] = 1
    if r.random() < .3: o["groupshift"] = r.choice([7, 8, 9])
    if r.random() < .25: o["localtree"] = r.choice([1, 2])
    if r.random() < .25 and "palette" not in o: o["localrct"] = r.randrange(42)
    if r.random() < .25 and "palette" not in o and o.get("tree") != 3: o["localpalette"] = r.choice([1, 2, 3])
    if r.random() < .2: o["passes"] = r.choice([2, 3])
    elif r.random() < .2: o["permute"] = 1
    if r.random() < .2: o["extra"] = r.choice([1, 2, 3])
    if r.random() < .3: o["alpha"] = 1
    elif r.random() < .2: o["bpp"] = r.choice([9, 10, 12, 14])
    if r.random() < .1: o["xyb"] = 1
    return o


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    flip_share = float(os.environ.get("FUZZ_FLIPS", "0.3"))
    cut_share = float(os.environ.get("FUZZ_CUTS", "0.1"))
    flip_from = float(os.environ.get("FUZZ_FROM", "0.33"))   # flips land in [this share of the file, end): 0 includes the headers
    flip_to = int(os.environ.get("FUZZ_TO", str(1 << 62)))           # ... or in the first FUZZ_TO bytes
    if on_gpu:
        import torch   # (loads the HIP runtime the library is to share)
        import j40_amd
    ref = Ref()
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    # FUZZ_SIMMODE: hostsim_decode's entropy path for VarDCT frames (tests/hostsim/hostsim.cpp): 4 = the throughput kernel's lanes,
    # 12 = one lane taking every section from a queue (k_hf_lanes' queued form), 16 = the latency kernel's fast path
    sim_mode = int(os.environ.get("FUZZ_SIMMODE", "0"))
    bad = skipped = errors = crashed = special = 0
    for i in range(n):
        mode = r.choice(["vardct", "modular"])
        w, h = r.randrange(260 if mode == "vardct" else 9, int(os.environ.get("FUZZ_MAXW", "900"))), r.randrange(8, int(os.environ.get("FUZZ_MAXH", "700")))   # (FUZZ_MAXW > 2048: several LF groups)
        if mode == "vardct" and w * h < 257 * 8 * 2: h = 264
        o = pick_vardct(r) if mode == "vardct" else pick_modular(r)
        if mode == "vardct" and os.environ.get("FUZZ_FORCE"):   # e.g. FUZZ_FORCE=maxlog=8: every VarDCT stream with 128 / 256-sized transforms
            for kv in os.environ["FUZZ_FORCE"].split(","):
                k, v = kv.split("=")
                o[k] = int(v)
            if "maxlog" in o and o["maxlog"] > 6: o.pop("forward", None); o.pop("detail", None); o.pop("beta", None)
        seed = r.randrange(1 << 20)
        try:
            d = synth(mode, w, h, seed, **o)
        except Exception as e:   # option combinations the generator refuses
            skipped += 1
            continue
        damaged = False
        if mode == "modular" and o.get("tree") != 5 and os.environ.get("FUZZ_SQUEEZE") and r.random() < float(os.environ["FUZZ_SQUEEZE"]):   # (tree 5's leaves quantise: no round trip)
            # Squeeze round trip (the reference stops at it with TODO): the squeezed stream must decode to what the reference makes
            # of the same picture coded without Squeeze. Clean streams only.
            try:
                sq = synth(mode, w, h, seed, squeeze=r.choice([1, 2, 3]), **o)
            except Exception:
                skipped += 1
                continue
            e, px = ref.decode(d)
            out = np.zeros((h, w, 4), np.uint8)
            buf = C.create_string_buffer(sq, len(sq))
            code = S.hostsim_decode(buf, len(sq), out.ctypes.data, None, 0)
            if e != "" or code != 0 or not np.array_equal(px, out):
                if e == "":   # (pictures the reference refuses even without Squeeze say nothing)
                    bad += 1
                    print("SQUEEZE MISMATCH", mode, w, h, seed, o, repr(e), code)
                    open("/tmp/fuzz_mismatch_%s_%d.jxl" % (sys.argv[2] if len(sys.argv) > 2 else "1", bad), "wb").write(sq)
            continue
        if r.random() < flip_share:   # one flipped bit somewhere behind the headers: the error code must match, too
            b = bytearray(d)   # (not in the size header: the buffers here are sized from the clean stream)
            b[r.randrange(max(12, int(len(d) * flip_from)), min(len(d), flip_to))] ^= 1 << r.randrange(8)
            d = bytes(b); damaged = True
        clean_len = len(d)
        if r.random() < cut_share:   # a truncated file, or junk behind it
            d = d[:r.randrange(20, len(d))] if r.random() < .7 else d + bytes(r.randrange(256) for _ in range(r.randrange(1, 40)))
        if os.environ.get("FUZZ_VERBOSE"): print(i, mode, w, h, seed, o, len(d), clean_len, flush=True); open("/tmp/fuzz_last.jxl", "wb").write(d)
        e, px = decode_in_child(ref, d) if (len(d) != clean_len or damaged) and not on_gpu else ref.decode(d)   # (no fork once HIP is up)
        if e == "CRSH":
            crashed += 1
            print("the reference crashed on", mode, w, h, seed, o, "(damaged stream)")
            continue
        if on_gpu:
            mine, out = j40_amd.decode(d)
            if out is None: out = np.zeros((h, w, 4), np.uint8)
        else:
            out = np.zeros((h, w, 4), np.uint8)
            buf = C.create_string_buffer(d, len(d))
            # (the lanes of modes 4 / 12 decode coefficients only; the harness leaves the Modular sub-images behind them -- extra
            # channels -- unchecked in those modes, as such frames never reach the lanes in the product: the general path for them)
            use_mode = 0 if (sim_mode & 4) and "alpha" in o else sim_mode
            code = S.hostsim_decode(buf, len(d), out.ctypes.data, None, use_mode)
            if use_mode and code == int.from_bytes(b"TODO", "big"):   # the stream is not one the special entropy path takes: the general one
                code = S.hostsim_decode(buf, len(d), out.ctypes.data, None, 0)
            elif use_mode and mode == "vardct": special += 1
            mine = "" if code == 0 else code.to_bytes(4, "big").decode("latin1")
        errors += e != ""
        if not on_gpu and len(d) > clean_len and e in ("excs", "shrt"):
            mine = e   # bytes behind the frame (and behind a container's last box) are judged by the public API
                       # (j40hip_frame_after_frame_status, api.cpp), not by this harness -- also when a section fails as well
        if e == "TODO": mine = e   # the reference stops at features it does not implement (a flipped bit can announce a Squeeze,
                                               # j40.h:3812, which this decoder carries out): whatever follows is not comparable
        ok = mine == e and (e != "" or (np.array_equal(px, out) if mode == "modular" else np.abs(px.astype(int) - out).max() <= 1))
        if not ok:
            bad += 1
            print("MISMATCH", mode, w, h, seed, o, repr(e), repr(mine))
            open("/tmp/fuzz_mismatch_%s_%d.jxl" % (sys.argv[2] if len(sys.argv) > 2 else "1", bad), "wb").write(d)   # (the stream as decoded, damage included)
    print("%d cases (%d refused by the generator, %d that the reference rejects, %d that crash it%s), %d mismatches" % (n, skipped, errors, crashed, ", %d through entropy path %d" % (special, sim_mode) if sim_mode else "", bad))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own kernels.

TEST INFRASTRUCTURE.  Runs only in the build container: it needs
oracle/_ref/libfosphor_ref.so, which oracle/Makefile (`make ref`) compiles from
/root/reference/lib/fosphor/{fft.cl,display.cl} where they lie.  The fixtures are data
(inputs + expected outputs); no reference source is stored.

    python3 oracle/gen_golden.py            # regenerate everything
    python3 oracle/gen_golden.py c1_gauss_b16

Every fixture is produced with the "portable" built-in binding
(include/fosphor_portable_math.h).  golden_meta.json additionally records, per case, how
many hit-count cells / waterfall texels change when the reference kernels are bound to
glibc's sinf/cosf/hypotf/log10f/roundf instead (informational: the libm sensitivity of an
implementation-defined OpenCL runtime).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases as gc			# noqa: E402
from oracle_lib import RefKernels, build_oracle, canon_bits, digest, hitcount_from_rows	# noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return digest(a)


def written_rows(wf, pos0, batch):
    idx = (pos0 + np.arange(min(batch, 1024))) & 1023
    if batch > 1024:		# ring overwritten: last 1024 spectra survive (display.cl:142-146)
        idx = (pos0 + np.arange(batch - 1024, batch)) & 1023
    return idx, wf[idx]


def run_case(name, spec, portable=True):
    r = RefKernels(portable=portable)
    if "power_range" in spec:
        r.set_power_range(*spec["power_range"])
    if "window" in spec:
        r.set_window(spec["window"]())
    out = []
    for x in spec["calls"]():
        pos0 = r.waterfall_pos
        rv = r.process(x, strict=spec.get("strict", True))
        assert rv == 0, (name, rv)
        batch = x.shape[0] // 1024
        wf = r.waterfall
        idx, rows = written_rows(wf, pos0, batch)
        # counts: every spectrum of THIS call, through its pwr row.  For batch <= 1024 all rows
        # survive in the ring; for larger batches take them from the FFT output instead.
        out.append(dict(x=x, batch=batch, pos0=pos0, pos1=r.waterfall_pos, fft=r.fft_out,
                        wf_idx=idx, wf_rows=rows, wf=wf, hist=r.histogram, spec=r.spectrum,
                        hs=r.histo_scale, ho=r.histo_offset))
    return out


def main(argv):
    build_oracle(ref=True)
    os.makedirs(OUT, exist_ok=True)
    meta_path = os.path.join(OUT, "golden_meta.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    names = [a for a in argv if a in gc.CASES] if argv else list(gc.CASES)

    for name in names:
        spec = gc.CASES[name]
        t0 = time.time()
        res = run_case(name, spec, portable=True)
        alt = run_case(name, spec, portable=False) if res[0]["batch"] <= 1024 else None
        arrays, m = {}, {"calls": []}
        for k, c in enumerate(res):
            pre = "c%d_" % k
            if c["batch"] <= 1024:
                hc = hitcount_from_rows(c["wf_rows"], c["hs"], c["ho"], 128)
            else:
                hc = None
            cm = dict(batch=c["batch"], pos0=c["pos0"], pos1=c["pos1"],
                      hs=float(c["hs"]), ho=float(c["ho"]),
                      sha_fft=sha(c["fft"]), sha_wf=sha(c["wf"]), sha_hist=sha(c["hist"]),
                      sha_spec=sha(c["spec"]))
            if hc is not None:
                cm["sha_hc"] = sha(hc)
                assert int(hc.sum()) == c["batch"] * 1024
            if alt is not None:
                a = alt[k]
                hca = hitcount_from_rows(a["wf_rows"], a["hs"], a["ho"], 128)
                cm["glibc_binding"] = dict(
                    hitcount_cells_differ=int((hca != hc).sum()),
                    waterfall_texels_differ=int((canon_bits(a["wf_rows"]) != canon_bits(c["wf_rows"])).sum()),
                    fft_words_differ=int((canon_bits(a["fft"]) != canon_bits(c["fft"])).sum()),
                    samples=int(c["batch"] * 1024))
            m["calls"].append(cm)
            if spec["store"] == "full":
                arrays[pre + "x"] = c["x"]
                arrays[pre + "fft"] = c["fft"]
                arrays[pre + "wf_idx"] = c["wf_idx"].astype(np.int32)
                arrays[pre + "wf_rows"] = c["wf_rows"]
                arrays[pre + "hist"] = c["hist"]
                arrays[pre + "spec"] = c["spec"]
                arrays[pre + "hc"] = hc
            elif k == len(res) - 1:
                arrays[pre + "hist"] = c["hist"]
                arrays[pre + "spec"] = c["spec"]
                arrays[pre + "wf_row_sample_idx"] = c["wf_idx"][::97].astype(np.int32)
                arrays[pre + "wf_row_sample"] = c["wf_rows"][::97]
                if hc is not None:
                    arrays[pre + "hc"] = hc
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        meta[name] = m
        print("%-22s %d call(s)  %.1fs" % (name, len(res), time.time() - t0), flush=True)

    if not argv or "glibc" in argv:
        # The same reference kernels with the math built-ins bound to glibc's sinf/cosf/hypotf/log10f/roundf
        # ("what a typical CPU OpenCL runtime would do") instead of the pinned portable header: the hit counts
        # of every call, for an INFORMATIONAL comparison with a stated cell budget (the built-ins are
        # implementation-defined in OpenCL, so neither binding is "the" reference).
        arrays = {}
        for name in ("c1_gauss_b16", "c2_tone_b32x3", "c5_wrap_b512_b1024", "c6_range_m20_5"):
            for k, c in enumerate(run_case(name, gc.CASES[name], portable=False)):
                arrays["%s_c%d_hc" % (name, k)] = hitcount_from_rows(c["wf_rows"], c["hs"], c["ho"], 128).astype(np.uint16)
        np.savez_compressed(os.path.join(OUT, "glibc_binding_hc.npz"), **arrays)
        meta["glibc_binding_hc"] = {k: sha(v) for k, v in arrays.items()}

    if not argv or "fft512" in argv:
        x = gc.fft512_input()
        y = RefKernels.fft(x, gc.hann512(), n=512)
        np.savez_compressed(os.path.join(OUT, "fft512.npz"), x=x, win=gc.hann512(), fft=y)
        meta["fft512"] = {"sha_fft": sha(y)}

    if not argv or "cmap" in argv:
        # palettes straight from the reference's gl_cmap_gen.c (oracle/_ref/libcmap_ref.so)
        import ctypes as C
        L = C.CDLL(os.path.join(HERE, "_ref", "libcmap_ref.so"))
        arrays = {}
        for fn in ("histogram", "waterfall", "prog"):
            f = getattr(L, "fosphor_gl_cmap_" + fn)
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
            for n in (256, 64, 1000):		# 256 is what gl.c:265-268 asks for
                buf = np.zeros(n, np.uint32)
                assert f(buf.ctypes.data, n, None) == 0
                arrays["%s_%d" % (fn, n)] = buf
        np.savez_compressed(os.path.join(OUT, "cmap_palettes.npz"), **arrays)
        meta["cmap_palettes"] = {k: sha(v) for k, v in arrays.items()}

    if not argv or "axis" in argv:
        # frequency-axis labels straight from the reference's axis.c (oracle/_ref/libaxis_ref.so)
        import ctypes as C

        class FreqAxis(C.Structure):		# axis.h:20-30
            _fields_ = [("center", C.c_double), ("span", C.c_double), ("step", C.c_double), ("mode", C.c_int),
                        ("abs_fmt", C.c_char * 16), ("abs_scale", C.c_double), ("rel_fmt", C.c_char * 16),
                        ("rel_step", C.c_double)]
        L = C.CDLL(os.path.join(HERE, "_ref", "libaxis_ref.so"))
        L.freq_axis_build.argtypes = [C.POINTER(FreqAxis), C.c_double, C.c_double, C.c_int]
        L.freq_axis_render.argtypes = [C.POINTER(FreqAxis), C.c_char_p, C.c_int]
        cases = []
        for center, span, n_div in gc.AXIS_CASES:
            fx = FreqAxis()
            L.freq_axis_build(C.byref(fx), center, span, n_div)
            labels = []
            for step in range(-(n_div // 2), n_div - n_div // 2 + 1):
                buf = C.create_string_buffer(64)
                L.freq_axis_render(C.byref(fx), buf, step)
                labels.append(buf.value.decode())
            cases.append(dict(center=center, span=span, n_div=n_div, mode=fx.mode,
                              abs_fmt=fx.abs_fmt.decode() if center != 0.0 else None,
                              abs_scale=fx.abs_scale if center != 0.0 else None,
                              rel_fmt=fx.rel_fmt.decode() if fx.mode == 1 else None,
                              rel_step=fx.rel_step if fx.mode == 1 else None, labels=labels))
        json.dump(cases, open(os.path.join(OUT, "axis_labels.json"), "w"), indent=1)

    if not argv or "geom" in argv:
        # render geometry / pixel <-> unit mapping straight from the reference's fosphor.c (oracle/_ref/geom_dump)
        import subprocess
        out = subprocess.check_output([os.path.join(HERE, "_ref", "geom_dump")]).decode()
        cases = json.loads(out)
        json.dump(cases, open(os.path.join(OUT, "render_geometry.json"), "w"), indent=1)

    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if a in gc.CASES or a in ("fft512", "cmap", "axis", "geom", "glibc")])

#!/usr/bin/env python
"""BASELINE configs[0] as a kept artefact: qoibench.c's measurement (qoibench.c:364-417: verification round trip, BENCHMARK_FN
timing, the table of :335-360) on ONE 512 x 512 RGBA PNG, with the unmodified reference (oracle/_ref) in the `qoi-ref` row and -
where a GPU is present - the MI355X library in the `qoi-mi355x` (host pointers) and `qoi-dev` (device-resident) rows.
usage: python tools/config0_table.py OUT.txt [runs]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    out_path = sys.argv[1]
    runs = sys.argv[2] if len(sys.argv) > 2 else "20"
    import png_io
    import qoibench_mi355x as qb
    from qoi_amd import synth
    use_gpu = False
    try:
        import torch
        use_gpu = torch.cuda.is_available()
    except Exception:
        pass
    d = tempfile.mkdtemp(prefix="qoi_cfg0_")
    px = synth.frame_rgba("photo", 512, 512, 3).reshape(512, 512, 4)
    open(os.path.join(d, "plumbing_512x512.png"), "wb").write(png_io.write_png(px))
    ref = os.path.join(ROOT, "oracle", "_ref", "libqoiref.so")
    lines = [f"# python tools/config0_table.py (tools/qoibench_mi355x.py {runs} <dir with one 512x512 RGBA PNG, synthetic 'photo' frame 3> "
             f"--ref-lib oracle/_ref/libqoiref.so --ref-prefix ref_); GPU rows: {'yes' if use_gpu else 'no GPU in this run'}"]
    args = [runs, d]
    if os.path.exists(ref):
        args += ["--ref-lib", ref, "--ref-prefix", "ref_"]
    rc = qb.main(args, out=lambda s="": lines.extend(str(s).split("\n")), use_gpu=use_gpu)
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    return rc


if __name__ == "__main__":
    sys.exit(main())

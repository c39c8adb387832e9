"""A/B builds of the demodulator for same-box timing: python profiles/probe/build_ab.py NAME[:-DFLAG[,-DFLAG...]] ...
Each NAME becomes acarsdec_amd/lib/ab/libNAME.so = the product objects with msk.hip recompiled under the given flags
(a git revision may be given as NAME@REV:flags to take msk.hip from history).  profiles/probe/run_ab.sh times them all
on one box through ACARSDEC_AMD_LIB.  Measurement aid only; nothing here is loaded by the product."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from acarsdec_amd import _build as B

MSK_FLAGS = B.MSK_FLAGS                 # the product's recipe (acarsdec_amd/_build.py)
B.build_lib()
out = os.path.join(B.LIBDIR, "ab")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    name, _, rev = name.partition("@")
    flags = [f for f in flags.split(",") if f]
    drop = [f[5:] for f in flags if f.startswith("DROP=")]          # DROP=substr: leave out the product flags containing it
    flags = [f for f in flags if not f.startswith("DROP=")]
    base = []
    it = iter(MSK_FLAGS)
    for f in it:
        if f == "-mllvm":
            v = next(it)
            if not any(d in v for d in drop):
                base += [f, v]
        elif not any(d in f for d in drop):
            base.append(f)
    src = os.path.join(B.CSRC, "msk.hip")
    if rev:
        src = os.path.join(out, "msk_%s.hip" % name)
        open(src, "w").write(subprocess.run(["git", "show", "%s:acarsdec_amd/csrc/msk.hip" % rev], cwd=ROOT, capture_output=True, text=True, check=True).stdout)
    obj = os.path.join(out, "msk_%s.o" % name)
    B._run([B.hipcc(), "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + B.INC, "-I" + B.CSRC] + base + flags + ["-c", src, "-o", obj])
    # msk_lean.hip (round 6) under the same flags: its A/B switches are ACG_LEAN_AB_*
    obj2 = os.path.join(out, "msk_lean_%s.o" % name)
    lean_base = [f for i, f in enumerate(base) if not (f == "-disable-machine-sink" or (f == "-mllvm" and i + 1 < len(base) and base[i + 1] == "-disable-machine-sink"))]
    B._run([B.hipcc(), "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + B.INC, "-I" + B.CSRC] + lean_base + flags + ["-c", os.path.join(B.CSRC, "msk_lean.hip"), "-o", obj2])
    objs = [os.path.join(B.OBJDIR, n) for n in ("fir.hip.o", "fir_mm.hip.o", "synth.hip.o", "blk.hip.o", "acg_api.cpp.o", "host_setup.o")] + [obj, obj2]
    lib = os.path.join(out, "lib%s.so" % name)
    B._run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-o", lib] + objs + ["-ldl", "-lm"])
    print(lib)

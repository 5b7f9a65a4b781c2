"""Print the register / scratch / LDS footprint of every kernel in the built code objects (llvm-readelf --notes metadata).
usage: python3 tools/kernel_resources.py [object or .so ...]   (default: build/*.o)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def notes(path):
    # host objects carry the device code as an offload bundle: unbundle the gfx950 part first
    tmp, fat = "/tmp/hxr_kres_%d.co" % os.getpid(), "/tmp/hxr_kres_%d.fatbin" % os.getpid()
    subprocess.run([OBJCOPY, "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True, text=True)
    r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + tmp],
                       capture_output=True, text=True)
    target = tmp if r.returncode == 0 and os.path.exists(tmp) and os.path.getsize(tmp) > 0 else path
    out = subprocess.run([READELF, "--notes", target], capture_output=True, text=True).stdout
    for f in (tmp, fat):
        if os.path.exists(f):
            os.remove(f)
    return out


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "*.o")))
    print("%-92s %5s %5s %6s %6s %8s %6s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratchB", "ldsB"))
    for f in files:
        txt = notes(f)
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = blk.split("amdhsa.target")[0]
            def g(k):
                m = re.search(r"\." + k + r":\s+(\S+)", blk)
                return m.group(1) if m else "?"
            m = re.search(r"\n\s+\.name:\s+(\S+)", blk)   # the kernel's own .name (arguments have none here: they are unnamed)
            name = demangle(m.group(1) if m else "?")
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            print("%-92s %5s %5s %6s %6s %8s %6s" % (name[:92], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                                                  g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()

"""Build-time lint of the compiled kernels (round 4: the cause of round 3's GPU memory fault).

The fault of fuzz seed 24 was a code-generation defect: in hxr_render_kernel<false, 239u, 3, -1> built with
-amdgpu-sched-strategy=max-memory-clause, the register allocator's live-range-split copy `v_mov_b32 v66, v100` (cur_shader saved while the
node scan reuses v100) was emitted as the FIRST instruction of a join block, AHEAD of the `s_or_b64 exec, exec, s[2:3]` that ends the
divergent region before it -- so it ran under that region's narrowed mask (or none), while the copy back ran under the full mask: lanes that
skipped the region got a stale shader index and loaded a shader record from a wild address.  (rocgdb: profiles/r04_fault_seed24_rocgdb.txt.)

This script disassembles every gfx950 kernel of the given objects and reports each vector instruction (VALU write / VMEM / LDS access) that sits
between a branch-target label and the exec-restoring `s_or_b64 exec, exec, ...` of that block: such an instruction executes under the mask of
the region that is being closed, which is never what a live-range split or a spill reload of a value that outlives the region needs.
Exit status 1 when anything is found.   usage: python3 tools/check_exec_prologue.py [objects ...]   (default: build/*.o)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(path):
    tmp, fat = "/tmp/hxr_lint_%d.co" % os.getpid(), "/tmp/hxr_lint_%d.fatbin" % os.getpid()
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True)
    r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + tmp], capture_output=True)
    target = tmp if r.returncode == 0 and os.path.exists(tmp) and os.path.getsize(tmp) > 0 else path
    out = subprocess.run([LLVM + "llvm-objdump", "-d", target], capture_output=True, text=True).stdout
    for f in (tmp, fat):
        if os.path.exists(f):
            os.remove(f)
    return out


def kernels(text):
    name, rows = None, []
    for l in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
        if m:
            if name and rows:
                yield name, rows
            name, rows = m.group(1), []
            continue
        m = re.search(r"//\s*([0-9A-F]{8,16}):", l)
        if m and name:
            rows.append((int(m.group(1), 16), l.split("//")[0].strip(), l))
    if name and rows:
        yield name, rows


# v_readlane / v_writelane / v_readfirstlane (SGPR spills and reloads) ignore the mask
VECTOR = re.compile(r"^(v_(?!readlane|writelane|readfirstlane)|global_|scratch_|flat_|buffer_(?!wbl2|inv)|ds_)")


def check(name, rows):
    base = rows[0][0]
    index_of = {addr: k for k, (addr, _, _) in enumerate(rows)}
    found = []
    for b, (addr, text, raw) in enumerate(rows):
        if not text.startswith("s_cbranch_execz"):
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", raw)
        if not m or base + int(m.group(1), 16) not in index_of:
            continue
        k = index_of[base + int(m.group(1), 16)]
        if k <= b:
            continue   # a loop's back edge, not the skip of a divergent region
        # the mask this region saved: the s_*_saveexec_b64 just before the skip branch
        saved = None
        for j in range(b - 1, max(b - 4, -1), -1):
            ms = re.match(r"s_\w+_saveexec_b64 (s\[\d+:\d+\]|vcc)", rows[j][1])
            if ms:
                saved = ms.group(1)
                break
        if not saved:
            continue
        # head of the join block: everything from the label to the restore of THAT mask must be scalar
        seen = []
        for j in range(k, min(k + 40, len(rows))):
            t = rows[j][1]
            if t == "s_or_b64 exec, exec, " + saved:
                found += [(rows[i][0], rows[i][1], t) for i in seen]
                break
            if re.match(r"s_(c?branch|setpc|endpgm|barrier)", t) or re.search(r"\bexec\b", t.split(",")[0]):
                break
            if re.match(r"(s_|v_readlane|v_writelane)", t) and re.search(r"\b" + re.escape(saved.split("[")[0]) + r"\b", t.split(",")[0]) and saved in t.split(",")[0]:
                break   # the saved mask's register pair is redefined before any restore: not this region's join
            if VECTOR.match(t):
                seen.append(j)
    return found


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "*.o")))
    bad = 0
    for f in files:
        for name, rows in kernels(disassemble(f)):
            hits = check(name, rows)
            if hits:
                short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
                for addr, text, restore in hits:
                    print("%s: %s +0x%x: `%s` runs ahead of `%s`" % (os.path.basename(f), short, addr - rows[0][0], text, restore))
                bad += len(hits)
    print("check_exec_prologue: %d vector instruction(s) ahead of an exec restore in %d object(s)" % (bad, len(files)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

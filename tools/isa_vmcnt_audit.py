"""ISA audit for the store -> load serialisation of gfx9's shared vmcnt: on gfx950 stores count in vmcnt like loads, so an
`s_waitcnt vmcnt(0)` in front of a loaded value also waits for every store issued before it.  A loop that issues its loads behind
the previous trip's stores (or any kernel with a branch between a pending load and its use, where hipcc falls back to vmcnt(0))
pays a store round trip per wait.  This script compiles a .hip file to assembly if needed and lists, per kernel, the loops that
contain loads, stores and a vmcnt(0): (instructions, loads, stores, vmcnt(0) waits) per trip -- many waits per trip, or one load
per wait, mark the candidates (round 4: ctc_rows_grad 113 -> 83 us, ctc_scan, convert.hip's tile).
  python tools/isa_vmcnt_audit.py wav2letter_amd/csrc/elementwise.hip [more .hip or .s files] [-DW2L_PROBE]"""
import os, tempfile
import re, sys, subprocess
# For each kernel: find loop bodies (from a "Loop Header" label to the last backward branch to it) and report loops where a
# global/buffer store is followed (cyclically) by a load whose use waits vmcnt(0): store -> load serialisation.
def demangle(n):
    try: return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
defs = [a for a in sys.argv[1:] if a.startswith('-D')]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in [a for a in sys.argv[1:] if not a.startswith('-D')]:
    if not f.endswith('.s'):
        out = os.path.join(tempfile.gettempdir(), os.path.basename(f) + '.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-S', '--cuda-device-only', *defs, '-I' + root + '/include',
                        '-I' + root + '/wav2letter_amd/csrc', '-o', out, f], check=True, stderr=subprocess.DEVNULL)
        f = out
    lines = open(f).read().split('\n')
    kern = None; start = {}
    i = 0
    funcs = []
    cur = None
    for ln, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m: cur = [m.group(1), ln, None]; funcs.append(cur)
        if l.startswith('.Lfunc_end') and cur: cur[2] = ln
    for name, a, b in funcs:
        if b is None: continue
        body = lines[a:b]
        labels = {}
        for k, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m: labels[m.group(1)] = k
        loops = []
        for k, l in enumerate(body):
            m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        rep = []
        for (s, e) in loops:
            seg = body[s:e]
            st = [k for k, l in enumerate(seg) if re.search(r'(global|buffer)_store', l)]
            ld = [k for k, l in enumerate(seg) if re.search(r'(global|buffer)_load', l)]
            w0 = [k for k, l in enumerate(seg) if 's_waitcnt vmcnt(0)' in l or re.search(r's_waitcnt\s+lgkmcnt\(\d+\)\s*$', l) and False]
            if st and ld and w0:
                n = len([l for l in seg if l.startswith('\t') and not l.strip().startswith(';')])
                rep.append((n, len(ld), len(st), len(w0)))
        if rep:
            print(f, demangle(name)[:90])
            for r in rep: print('    loop: %d instrs, %d loads, %d stores, %d vmcnt(0)' % r)

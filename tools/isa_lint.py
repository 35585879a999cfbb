"""ISA lint of the SHIPPED artifact: disassembles every gfx950 code object inside a built library and fails on the one
packed-fp32 instruction form gfx950 misreads next to another kernel's 16-bit MFMA waves -- v_pk_{add,mul,fma}_f32 whose
LOW result reads the HIGH half of a VGPR src1, op_sel:[x,1,...] (tools/micro/corun6.hip).

    python tools/isa_lint.py nisqa_amd/libnisqa_hip.so [ARCH]    (run by the Makefile after linking; exit code 1 on a match)

ARCH (default gfx950) is the --offload-arch of the build: the finding is specific to gfx950, so a library built for
another architecture is left alone with a notice.  A missing llvm-objdump FAILS the build unless NISQA_SKIP_LINT=1 says
the skip is intended (tests/test_host.py lints the sources either way).
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

BAD = re.compile(r'v_pk_(add|mul|fma)_f32\b.*\bop_sel:\[[01],1')
LLVM = os.environ.get('NISQA_LLVM_BIN', '/opt/rocm/lib/llvm/bin')


def scan_library(path):
    """-> (number of code objects, number of packed-f32 instructions seen, list of offending lines)"""
    objdump = os.path.join(LLVM, 'llvm-objdump')
    if not os.path.isfile(objdump):
        raise RuntimeError('llvm-objdump not found under ' + LLVM)
    tmp = tempfile.mkdtemp(prefix='nq_lint_')
    try:
        lib = os.path.join(tmp, os.path.basename(path))
        shutil.copyfile(path, lib)
        subprocess.run([objdump, '--offloading', lib], cwd=tmp, check=True, capture_output=True)
        objs = sorted(glob.glob(lib + '.*gfx950*'))
        n_pk, bad = 0, []
        for o in objs:
            txt = subprocess.run([objdump, '-d', o], check=True, capture_output=True, text=True).stdout
            for line in txt.split('\n'):
                if 'v_pk_' in line and '_f32' in line:
                    n_pk += 1
                    if BAD.search(line):
                        bad.append(os.path.basename(o) + ': ' + line.strip())
        return len(objs), n_pk, bad
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    path = sys.argv[1]
    arch = sys.argv[2] if len(sys.argv) > 2 else 'gfx950'
    if not arch.startswith('gfx950'):
        print('isa_lint: built for %s, not gfx950 -- the packed-f32 op_sel finding does not apply, nothing linted' % arch)
        return 0
    if os.environ.get('NISQA_SKIP_LINT') == '1':
        print('isa_lint: skipped (NISQA_SKIP_LINT=1)')
        return 0
    if not os.path.isfile(os.path.join(LLVM, 'llvm-objdump')):
        print('isa_lint: llvm-objdump not found under %s (NISQA_LLVM_BIN) -- the artifact cannot be linted; set '
              'NISQA_SKIP_LINT=1 to build without the lint' % LLVM)
        return 1
    n_obj, n_pk, bad = scan_library(path)
    if n_obj == 0 or n_pk == 0:
        print('isa_lint: found %d gfx950 code objects and %d packed-f32 instructions in %s: the scan itself is broken' % (n_obj, n_pk, path))
        return 1
    if bad:
        print('isa_lint: %d packed-f32 instruction(s) with op_sel on the low half of src1 in %s, e.g.\n  %s' % (len(bad), path, bad[0]))
        return 1
    print('isa_lint: %s clean (%d code objects, %d packed-f32 instructions)' % (path, n_obj, n_pk))
    return 0


if __name__ == '__main__':
    sys.exit(main())

"""Source rewriting for the CPU lane-level executor (tests/emu/include/emu_rt.h). TEST INFRASTRUCTURE.

The kernel sources under yolov5m_amd/csrc are compiled UNCHANGED for the emulator except for the few constructs a host
compiler cannot take, which this script rewrites textually (the product sources are never touched):
  * `extern __shared__ ... T name[];`            -> a pointer to the launch's dynamic LDS buffer
  * `asm volatile("s_waitcnt vmcnt(n)")`         -> emu_waitcnt_vm(n): the wave's LDS-DMA loads beyond the n newest land (they do
                                                    NOT land earlier: a missing / too-weak wait reads stale LDS), then a rendezvous
  * `asm volatile("s_waitcnt lgkmcnt(n)")`       -> emu_waitcnt_lgkm(n): a wave-level rendezvous (lanes run one after the other)
  * the LDS-DMA instruction of the halo kernel   -> emu_buffer_load_lds16(...)
  * `s_memtime` stamps                           -> 0
  * the "v" register constraint of empty asm statements (optimisation fences) -> "r"
Anything else that looks like GPU assembly is an error: a new asm statement needs a rule here."""
import re
import sys

RULES = [
    (re.compile(r'extern\s+__shared__\s+__attribute__\(\(aligned\(\d+\)\)\)\s+([\w ]+?)\s+(\w+)\[\];'),
     r'\1* \2 = reinterpret_cast<\1*>(emu::dyn_lds());'),
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)"\s*:::\s*"memory"\);'), r'emu_waitcnt_vm(\1);'),
    (re.compile(r'asm volatile\("s_waitcnt lgkmcnt\((\d+)\)"\s*:::\s*"memory"\);'), r'emu_waitcnt_lgkm(\1);'),
    (re.compile(r'asm volatile\("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds"\s*'
                r'::\s*"s"\((\w+)\),\s*"v"\((\w+)\),\s*"s"\((\w+)\),\s*"s"\((\w+)\)\s*:\s*"memory"\);', re.S),
     r'emu_buffer_load_lds16(\3, \2, \4, \1);'),
    (re.compile(r'asm volatile\("s_memtime %0\\n\\ts_waitcnt lgkmcnt\(0\)"\s*:\s*"=s"\((\w+)\)\s*::\s*"memory"\);'), r'\1 = 0;'),
    (re.compile(r'asm\(""\s*:\s*"\+v"\((\w+)\)\);'), r'asm("" : "+r"(\1));'),
]


def translate(text, name="<src>"):
    for pat, rep in RULES:
        text = pat.sub(rep, text)
    left = [m.group(0)[:80] for m in re.finditer(r'asm\s*(?:volatile)?\s*\("(?!")[^;]*;', text)]
    if left:
        raise RuntimeError(f"{name}: asm statement without an emulator rule: {left[0]!r}")
    return text


if __name__ == "__main__":
    src, dst = sys.argv[1:3]
    with open(src) as f:
        out = translate(f.read(), src)
    with open(dst, "w") as f:
        f.write(f'#line 1 "{src}"\n' + out)

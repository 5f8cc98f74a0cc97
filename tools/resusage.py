"""Per-kernel register / scratch / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc ... -Rpass-analysis=kernel-resource-usage -o lib.so adkf_gp.hip 2> remarks.txt
    python tools/resusage.py remarks.txt [substring ...]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
    rows = []
    for b in blocks:
        name = b.split('\n')[0].strip()

        def g(k):
            m = re.search(k + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        rows.append((name, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
                     g(r'LDS Size \[bytes/block\]')))
    dem = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
    for r, d in zip(rows, dem):
        d = re.sub(r'adkf::', '', d)
        d = re.sub(r'^void ', '', d)
        if pats and not any(p in d for p in pats):
            continue
        print(f"{d[:100]:100s} vgpr {r[1]:4d} agpr {r[2]:3d} sgpr {r[3]:4d} scratch {r[4]:5d} occ {r[5]:2d} lds {r[6]:7d}")


if __name__ == '__main__':
    main()

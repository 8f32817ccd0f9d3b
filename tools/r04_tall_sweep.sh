#!/bin/bash
# tools/r04_tall_sweep.sh: per-launch durations of one decode with the conv tile height by the heuristic and forced (ACE355_CONV_TM=128 / 256):
# after the Snake move the transposed and k = 1 convs stage plain rows - is the 4-wave tile still the better one for them?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_tall_sweep.txt
{
for v in 0 128 256; do
  echo "== ACE355_CONV_TM=$v (0: heuristic)"
  rm -rf /tmp/ts_$v
  ACE355_CONV_TM=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/ts_$v -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/ts_$v | tail -33
done
} > $OUT 2>&1
python - <<'PY'
import re
t = open('gpurun_out/r04_tall_sweep.txt').read().split('== ACE355_CONV_TM=')[1:]
names = ['conv1']
for C in [1024, 512, 256, 128, 128]:
    names.append(f'convT->{C}')
    for u in range(3):
        names += ([f'k7@{C}', f'k1@{C}'] if C >= 256 else [f'fused@{C}'])
names.append('out')
cols = {}
for sec in t:
    key = sec.split()[0]
    rows = [(float(m.group(1)), m.group(2)) for m in re.finditer(r'\s*([\d.]+) us grid=\S+ wg=(\d+) lds=\d+ conv_kernel', sec)]
    ends = [i for i, l in enumerate(re.findall(r'[\d.]+ us grid=\S+ wg=\d+ lds=\d+ (conv_kernel<[^>]*>)', sec)) if l.startswith('conv_kernel<32')]
    d = rows[ends[-1] + 1 - len(names):ends[-1] + 1]
    cols[key] = d
print('launch        ' + '   '.join(f'TM={k:>4s}' for k in cols))
for i, nm in enumerate(names):
    print(f'{nm:12s} ' + '  '.join(f'{cols[k][i][0]:8.1f}/{cols[k][i][1]}' for k in cols))
print('total        ' + '  '.join(f'{sum(x[0] for x in cols[k])/1e3:8.2f} ms ' for k in cols))
PY

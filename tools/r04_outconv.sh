#!/bin/bash
# tools/r04_outconv.sh: the 32-column conv kernel with every tap's weight tile of a chunk requested at once (product) against the
# two-tile ring (tools/_ab/lib_prev.so).  NOT KEPT (- 80 us on one launch): the change lives in tools/r04_outconv_alltaps.patch: sha of decode / encode, ABAB decode time, the output conv's launch time under the tracer.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_outconv.txt
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_new.so; cp tools/_ab/lib_prev.so /tmp/_old.so
{
for r in 1 2 3 4; do
  for v in old new; do
    cp /tmp/_$v.so $LIB
    echo "$v: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
for v in old new; do
  cp /tmp/_$v.so $LIB
  rm -rf /tmp/oc_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/oc_$v -- python tools/vae_trace.py > /dev/null 2>&1
  echo "$v, 32-column launches: $(python tools/vae_trace_list.py /tmp/oc_$v | grep '<32, 128>' | tr '\n' ' ')"
done
cp /tmp/_new.so $LIB
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -2
} > $OUT 2>&1
cp /tmp/_new.so $LIB
cat $OUT

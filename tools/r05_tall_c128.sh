cd /root/repo; export TMPDIR=/tmp
for tm in 128 256; do
  echo "== ACE355_CONV_FUSE_RU=0 ACE355_CONV_TM=$tm"
  rm -rf /tmp/ct_$tm
  ACE355_CONV_FUSE_RU=0 ACE355_CONV_TM=$tm rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_$tm -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/ct_$tm | tail -42
done

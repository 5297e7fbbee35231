#!/bin/bash
# The real shape on the one-GPU box: 512^3 over 8 ranks of 64 planes each (all sharing the device), direct halos through IPC
# mappings (ipc-host: only the 8-byte all-reduce is staged by the host) and the packed host transport, against the single domain:
# whole dumps (six fields of 512^3 + clock) byte for byte.  Proves rendezvous size, 32-bit offsets and memory at that shape.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out; T=/tmp/ring512; mkdir -p $T $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
G="--n 512 --frames 3 --start 1"
{
bin/tau3d $G --dump $T/single.bin | tail -3
for tr in ipc-host host; do
  bin/tau3d $G --gpus 8 --transport $tr --dump $T/w8.bin | tail -3
  if cmp -s $T/single.bin $T/w8.bin; then echo "512^3, 8 ranks x 64 planes, --transport $tr: dump ($(stat -c %s $T/w8.bin) bytes) IDENTICAL to the single domain"; else echo "512^3, 8 ranks, --transport $tr: dump DIFFERS"; fi
  rm -f $T/w8.bin
done
python bench.py --gpus 8 --ring-transport ipc-host --steps 4 --warmup 2 2>/dev/null | grep '^{' | cut -c1-1200
} > $O/ring_512_8ranks.txt 2>&1
rm -rf $T
cat $O/ring_512_8ranks.txt

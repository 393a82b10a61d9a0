#!/bin/bash
# mrc_tarx over 2 GiB files in /dev/shm (BASELINE config 5 shape on one GPU): wall time and GB/s of floats, writes on and -d 1,
# three runs each: bash tools/jobs/cli_tarx_2gib.sh [nfiles]      (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/datacompressionfloat_amd/bin
NF=${1:-4}
D=/dev/shm/mrcz_tarx_$$; mkdir -p $D/z $D/u $D/z2 $D/u2
trap "rm -rf $D" EXIT
python3 - $D $NF <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from datacompressionfloat_amd import MrcZipCodec
d, nf = sys.argv[1], int(sys.argv[2])
c = MrcZipCodec(0, max_batch_chunks=1)
fl = 1 << 29
names = []
for i in range(nf):
    w = torch.empty(fl, dtype=torch.int32, device="cuda")
    c.generate_kat_device(w, i * fl + 977 * i)
    p = f"{d}/vol{i}.mrc"; w.cpu().numpy().tofile(p); names.append(p)
open(f"{d}/files.txt", "w").write("\n".join(names) + "\n")
open(f"{d}/zips.txt", "w").write("\n".join(f"{d}/z/vol{i}.mrc.zip" for i in range(nf)) + "\n")
PY
BYTES=$((NF * 2147483648))
t() { local a=$(date +%s.%N); "$@" > $D/o.log 2> $D/e.err || { echo FAILED; tail -3 $D/e.err; }; local b=$(date +%s.%N); python3 -c "print('  wall %.3f s  %.2f GB/s' % ($b-$a, $BYTES/($b-$a)/1e9))"; }
for n in ${NS:-1 2 4}; do
  for rep in 1 2 3; do echo "zip -n $n run $rep:"; t $B/mrc_tarx -i $D/files.txt -t zip -o $D/z -b 8 -n $n; done
done
for n in ${NS:-1 2 4}; do
  for rep in 1 2 3; do echo "unzip -n $n run $rep:"; t $B/mrc_tarx -i $D/zips.txt -t unzip -o $D/u -n $n; done
done
cmp $D/u/vol0.mrc <($B/erasebytes -i $D/vol0.mrc -o /dev/stdout -b 8 2>/dev/null) > /dev/null 2>&1 && echo "decoded vol0 == erasebytes(vol0)"
for n in ${NSD:-2 4}; do
  for rep in 1 2 3; do echo "zip -d 1 -n $n run $rep:"; t $B/mrc_tarx -i $D/files.txt -t zip -o $D/z2 -b 8 -n $n -d 1; done
  for rep in 1 2 3; do echo "unzip -d 1 -n $n run $rep:"; t $B/mrc_tarx -i $D/zips.txt -t unzip -o $D/u2 -n $n -d 1; done
done
echo "trace of one zip and one unzip (-n 2):"
MRCZ_TRACE=1 $B/mrc_tarx -i $D/files.txt -t zip -o $D/z -b 8 -n 2 2>&1 | grep "mrcz trace" | head -12
MRCZ_TRACE=1 $B/mrc_tarx -i $D/zips.txt -t unzip -o $D/u -n 2 2>&1 | grep "mrcz trace" | head -12

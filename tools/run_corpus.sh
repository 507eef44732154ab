#!/bin/bash
# The reference's corpus workflow on the MI355X path: every .png of a directory through fpng_amd_test -c (1-pass and 2-pass),
# every output compared byte for byte with the UNMODIFIED reference encoder; one CSV line per file and mode.
#   tools/run_corpus.sh <dir> <out.csv>
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
DIR=$1; CSV=$2
JUDGE=$R/oracle/_ref/libfpng_ref.so; [ -f $JUDGE ] || JUDGE=$R/oracle/libfpng_oracle.so
echo "# fpng_amd_test -c [-s] --judge $(basename $JUDGE) -b 8 -p 1: file, w, h, chans, encode s (drop-in, PCIe incl.), size MiB, decode s, encode MiP/s, decode MiP/s, encode MP/s, host-batch s/frame, device-resident s/frame, CPU 1-thread s, CPU s/image" > $CSV
rc=0
for f in $DIR/*.png; do
  for m in "" "-s"; do
    LD_PRELOAD= $R/fpng_amd/lib/fpng_amd_test -c $m --judge $JUDGE -b 8 -p 1 -o /tmp/corpus_out.png $f >> $CSV || { echo "FAILED: $f $m" | tee -a $CSV; rc=1; }
  done
done
# alpha workflows of the harness: -a (green -> alpha) and an alpha source file
LD_PRELOAD= $R/fpng_amd/lib/fpng_amd_test -c -a --judge $JUDGE -o /tmp/corpus_out.png $DIR/photo_rgb.png >> $CSV || rc=1
LD_PRELOAD= $R/fpng_amd/lib/fpng_amd_test -c --judge $JUDGE -o /tmp/corpus_out.png $DIR/photo_rgb.png $DIR/photo_grey.png >> $CSV || rc=1
echo "corpus rc=$rc"
exit $rc

#!/bin/bash
# Joules per 256 x 256 x 64 tile step ("judge variants by joules per tile"): socket power (rocm-smi, 1 s apart) x time per
# launch / (tiles x K tiles) while ONE shape runs back to back, for the eight-wave product kernel (lab --sch=1), the four-wave lab kernels on
# the 32x32x16 (--sch=4) and 16x16x32 (--sch=16) MFMA, and the vendor library through torch (yardstick only).   usage (GPU box): bash tools/gemm_joules.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools/gemm_lab
sample() {   # $1 = pid to watch, prints "W sclk" averages over 4 samples
  sleep 4
  P=0; C=0; N=0
  for i in 1 2 3 4; do
    S=$(rocm-smi --showpower --showclocks 2>/dev/null)
    w=$(echo "$S" | grep -E "Socket" | grep -oE "[0-9]+\.[0-9]+" | head -1)
    c=$(echo "$S" | grep -E "sclk" | grep -oE "\([0-9]+Mhz\)" | grep -oE "[0-9]+" | head -1)
    [ -n "$w" ] && P=$(python3 -c "print($P+$w)") && C=$(python3 -c "print($C+${c:-0})") && N=$((N+1))
    sleep 1
  done
  python3 -c "n=max($N,1); print('%.0f W  %.0f MHz' % ($P/n, $C/n))"
}
for shape in sq4k bert_inter; do
  case $shape in sq4k) TILES=256; KT=64; NL=90000;; bert_inter) TILES=768; KT=12; NL=130000;; esac
  for sch in 1 4 16; do
    ./lab --n=$NL --plain-only --sch=$sch --quick --forms=1 $shape > /tmp/gj_$$.log 2>&1 &
    LP=$!
    PW=$(sample $LP)
    wait $LP
    US=$(grep "plain" /tmp/gj_$$.log | awk '{print $(NF-3)}' | tail -1)
    python3 -c "
w=float('$PW'.split()[0]); us=float('$US'); 
print('%-11s lab sch=%-2s  %8.1f us  $PW  %7.1f uJ per 256x256x64 tile step  (%.3f us per K tile per CU-round)' % ('$shape', '$sch', us, w*us/($TILES*$KT), us/($KT*max(1,$TILES/256))))"
  done
  python3 - <<PY > /tmp/gj_v_$$.log 2>&1 &
import torch, time
S={"sq4k":(4096,4096,4096),"bert_inter":(32768,1536,768)}["$shape"]
M,N,K=S; dev=torch.device("cuda:0"); torch.manual_seed(0)
x=torch.randn(M,K,device=dev).bfloat16(); w=(torch.randn(N,K,device=dev)*K**-0.5).bfloat16()
f=lambda: torch.nn.functional.linear(x,w)
for _ in range(10): f()
torch.cuda.synchronize(); n=$NL
a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): f()
e.record(); torch.cuda.synchronize()
print(a.elapsed_time(e)/n*1e3)
PY
  VP=$!
  PW=$(sample $VP)
  wait $VP
  US=$(tail -1 /tmp/gj_v_$$.log)
  python3 -c "
w=float('$PW'.split()[0]); us=float('$US');
print('%-11s vendor (torch)  %8.1f us  $PW  %7.1f uJ per 256x256x64 tile step  (%.3f us per K tile per CU-round)' % ('$shape', us, w*us/($TILES*$KT), us/($KT*max(1,$TILES/256))))"
done

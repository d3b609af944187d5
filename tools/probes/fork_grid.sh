# round 6: lanes x piece grid of the forked device-resident job, one process per (B, lanes)   usage: fork_grid.sh TAG "B..." "LANES..." "PIECES..."
O=gpurun_out/$1; mkdir -p $O
for B in $2; do for L in $3; do
  specs=""; for W in $4; do specs="$specs $L:dev_fork=1,dev_piece_waves=$W"; done
  timeout 300 python tools/dev_fork_rate.py $B $specs 2>>$O/err.log | tee -a $O/grid.jsonl
done; done

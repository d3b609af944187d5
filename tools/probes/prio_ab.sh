O=gpurun_out/r06e; mkdir -p $O; L=mina_bridge_amd/libminaverify.so
for v in prio0 prio2 prio3; do cp tools/probes/bin/lib_$v.so $L
  for B in 16384 8192 1024; do echo -n "$v "; timeout 300 python tools/dev_fork_rate.py $B "1:dev_fork=1,dev_piece_waves=3072" "1:dev_fork=1,dev_piece_waves=4096" "1:dev_fork=1" "1:dev_fork=1,dev_piece_waves=1024" "4:dev_fork=1,dev_piece_waves=1024" "4:dev_fork=1,dev_piece_waves=2048" "4:dev_fork=1" "20:dev_fork=0" 2>>$O/err.log | sed "s/^/$v /" | tee -a $O/prio_ab.jsonl; done; done
cp tools/probes/bin/lib_prio2.so $L

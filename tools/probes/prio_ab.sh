# round 6: same-box A/B of wave-priority builds (tools/probes/bin/lib_prio<chain><leg>.so: -DMB_CHAIN_PRIO / -DMB_LEG_PRIO) on the forked device-resident job
O=gpurun_out/${1:-r06p}; mkdir -p $O; L=mina_bridge_amd/libminaverify.so; cp $L /tmp/lib_keep.so
for rep in 1 2; do for v in ${VARIANTS:-prio22 prio31 prio32}; do cp tools/probes/bin/lib_$v.so $L
  for B in 16384 8192; do timeout 300 python tools/dev_fork_rate.py $B "1:dev_fork=1" "4:dev_fork=1" 2>>$O/err.log | sed "s/^/$v /" | tee -a $O/prio_ab.jsonl; done; done; done
cp /tmp/lib_keep.so $L

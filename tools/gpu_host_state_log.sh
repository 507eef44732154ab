#!/bin/bash
# runtime log (AMD_LOG_LEVEL=4) of the host state probe: which hardware queue each stream gets, which engine each copy takes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/host_state_log; rm -rf $O; mkdir -p $O
AMD_LOG_LEVEL=4 python tools/host_state_probe.py 2 > $O/out.txt 2> $O/full.log
tail -1 $O/out.txt
wc -l $O/full.log
grep -i -E "queue|engine|sdma|blit|stream" $O/full.log | grep -v -i "kernarg\|signal" | head -3000 > $O/sel.log
grep -o -i -E "(hsa_amd_memory_async_copy[a-z_]*|copy_engine[^,]*|acquire[A-Za-z]*|Selected queue[^,]*|created hardware queue[^,]*|hipStreamCreate[A-Za-z]*|ShaderName : [_a-zA-Z0-9]*copy[A-Za-z]*)" $O/full.log | sort | uniq -c | sort -rn | head -40
head -c 3000000 $O/full.log > $O/head.log
tail -c 6000000 $O/full.log > $O/tail.log
rm $O/full.log

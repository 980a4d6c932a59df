# A/B of the exchange variant of the path kernel against the default, scene1 1920x1080, 256 frames per launch,
# then the step statistics of both (instrumented kernels).  Run on the GPU box from the repository root.
X="MI355PT_XCHG=1;MI355PT_BLOCK=1024"
AB_F=256 AB_ROUNDS=5 timeout -k 10 250 python tools/ab.py "base=MI355PT_BLOCK=1024" \
 "exchange=$X" \
 "exchange_24_12=$X;MI355PT_XCHG_SCHED=12,8,24,24,12,32" \
 "exchange_32_32=$X;MI355PT_XCHG_SCHED=12,8,24,32,32,32" \
 "exchange_all=$X;MI355PT_XCHG_SCHED=12,8,24,65,65,32" \
 > gpurun_out/ab_xchg_final.log 2>&1 && \
PH_F=256 MI355PT_XCHG=1 MI355PT_BLOCK=1024 timeout -k 10 100 python tools/phases.py > gpurun_out/ph_xchg.log 2>&1 && \
PH_F=256 MI355PT_BLOCK=1024 timeout -k 10 100 python tools/phases.py > gpurun_out/ph_base.log 2>&1
tail -6 gpurun_out/ab_xchg_final.log; cat gpurun_out/ph_xchg.log gpurun_out/ph_base.log

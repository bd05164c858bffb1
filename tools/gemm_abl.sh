cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 4 8; do DH_G2_NSTAGE=2 DH_G2_ABL=$v python tools/gemm_bench.py gemm > gpurun_out/r3_ga_$v.txt 2>&1; done
paste gpurun_out/r3_ga_0.txt gpurun_out/r3_ga_1.txt gpurun_out/r3_ga_2.txt gpurun_out/r3_ga_3.txt gpurun_out/r3_ga_4.txt gpurun_out/r3_ga_8.txt | awk '{print $1,$2,$3,"full",$4,"noDMA",$12,"noST",$20,"noDMA+ST",$28,"noMFMA",$36,"empty",$44}'

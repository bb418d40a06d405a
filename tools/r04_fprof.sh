cd $GRAFT_REPO_ROOT
PBC_HIP_LIB=libpbc_hip_mo.so python tools/whatif_time.py f 18 4 > gpurun_out/r04_f_milleronly.txt 2>&1
python tools/whatif_time.py f 18 4 > gpurun_out/r04_f_full.txt 2>&1
bash tools/r03_pmc.sh f r04f > gpurun_out/r04_pmc_f.txt 2>&1
cat gpurun_out/r04_f_milleronly.txt gpurun_out/r04_f_full.txt; tail -5 gpurun_out/r04_pmc_f.txt

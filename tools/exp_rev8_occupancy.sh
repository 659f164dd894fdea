mkdir -p gpurun_out/r04c
o=gpurun_out/r04c/exp1.txt
run() { echo "== $*" >> $o; env "$@" python tools/rev8_run.py 125000 2 0 2>&1 | grep "walk ms" >> $o; }
run X=0
run MFA_WALK_IMAGES_GLOBAL=1
run MFA_WALK_C=2
run MFA_WALK_C=8 MFA_WALK_IMAGES_GLOBAL=1
run MFA_LIB_PATH=$PWD/build/libmfa_mw3.so MFA_WALK_WGS=3
run MFA_LIB_PATH=$PWD/build/libmfa_mw3.so MFA_WALK_WGS=3 MFA_WALK_IMAGES_GLOBAL=1
run MFA_LIB_PATH=$PWD/build/libmfa_mw4.so MFA_WALK_WGS=4
run MFA_LIB_PATH=$PWD/build/libmfa_mw4.so MFA_WALK_WGS=4 MFA_WALK_IMAGES_GLOBAL=1
run MFA_LIB_PATH=$PWD/build/libmfa_mw4.so MFA_WALK_WGS=2
cat $o

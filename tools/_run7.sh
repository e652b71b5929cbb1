for cfg in "3 8" "3 4"; do set -- $cfg
echo "== layout $1 nw $2"
GML_BWD_LAYOUT=$1 GML_BWD_NW=$2 GML_LIB=$PWD/_ab/lib_timing.so python3 tools/bwd2_phases.py 2>&1 | grep -v "amdgpu.ids"
done

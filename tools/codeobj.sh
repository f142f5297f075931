#!/bin/bash
# usage: tools/codeobj.sh <file.hip.o> [kernel-name-regex] -- registers, spills and scratch of the gfx950 kernels inside a host object
# (the metadata the code object carries; the extracted code object stays at /tmp/<name>.co for llvm-objdump -d)
o=$1; re=${2:-.}
b=$(basename $o)
d=$(mktemp -d)
cp $o $d/$b
( cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $b > /dev/null 2>&1 )
co=$(ls $d/$b.*gfx950* 2>/dev/null | head -1)
cp "$co" /tmp/${b%.o}.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/${b%.o}.co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|\.vgpr_spill|\.sgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - - | awk '{$1=$1};1' | grep -E "$re" | sed 's/\.private_segment_fixed_size/scratch/; s/\.group_segment_fixed_size/lds/'
rm -rf $d

#!/bin/bash
# Prove that a source change left the DEVICE code of an existing build untouched (used for the fp16 build, where the bf16 library
# had to stay byte-identical, and before every GPU-less default flip): disassemble every gfx950 code object embedded in two hipcc
# objects / shared libraries and compare.  hipcc objects themselves are not reproducible byte for byte (build ids), the device ISA is.
#   tools/device_code_diff.sh old.so new.so         -> "identical" / first differing lines
#   tools/device_code_diff.sh --meta file.o         -> per kernel: VGPRs, AGPRs, scratch bytes, spills, LDS
set -e
LLVM=/opt/rocm/lib/llvm/bin
extract() {   # $1 = object / library, $2 = output directory for the code objects
    mkdir -p "$2"; cp "$(realpath "$1")" "$2/x.o"; (cd "$2" && $LLVM/llvm-objdump --offloading x.o > /dev/null)
}
if [ "$1" = "--meta" ]; then
    D=$(mktemp -d); extract "$2" "$D"
    for f in "$D"/x.o.*gfx950; do
        $LLVM/llvm-readelf --notes "$f" | grep -E "\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed|group_segment_fixed|vgpr_spill" | paste - - - - - - | sed 's/  */ /g'
    done
    rm -rf "$D"; exit 0
fi
A=$(mktemp -d); B=$(mktemp -d)
extract "$1" "$A"; extract "$2" "$B"
for d in "$A" "$B"; do
    : > "$d/all.s"
    for f in $(ls "$d"/x.o.*gfx950 | sort -t. -k3 -n); do $LLVM/llvm-objdump -d "$f" | grep -v "file format" >> "$d/all.s"; done
done
if cmp -s "$A/all.s" "$B/all.s"; then
    echo "identical: $(ls "$A"/x.o.*gfx950 | wc -l) code objects, $(wc -l < "$A/all.s") lines of ISA"
else
    echo "DIFFERENT"; diff "$A/all.s" "$B/all.s" | head -20
fi
rm -rf "$A" "$B"

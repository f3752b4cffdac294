#!/bin/bash
# Does the WIDTH of the fused core's backward tile matter?    The row-space kernel (csrc/nmf_cf_gram.hip) with 4
# (shipped), 8 or 16 patches along W per workgroup — 128-, 256- and 512-byte runs per (channel, depth slice, row); 16 patches = the
# whole W row at 128^3, i.e. 4 KB contiguous per (channel, depth slice) — full kernel and memory skeleton (-DFZ_PROBE_GRAM_NOMATH).
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_wpb8.so  -DFZ_PROBE_GRAM_WPB=8 nmf_cf_gram.hip
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_wpb8n.so -DFZ_PROBE_GRAM_WPB=8,-DFZ_PROBE_GRAM_NOMATH nmf_cf_gram.hip
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_wpb16n.so -DFZ_PROBE_GRAM_WPB=16,-DFZ_PROBE_GRAM_NOMATH nmf_cf_gram.hip
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_wpb4n.so -DFZ_PROBE_GRAM_NOMATH nmf_cf_gram.hip
out=${1:-gpurun_out/r06/gram_tile.jsonl}
mkdir -p $(dirname $out); : > $out
for rep in 1 2; do
for v in shipped wpb8 wpb4n wpb8n wpb16n; do
  case $v in
    shipped) env= ;;
    *) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_$v.so" ;;
  esac
  env $env python tools/probes/gram_bwd.py time 2>gpurun_out/r06/gram_tile_$v.err | grep '"relu_gate": 1' | sed "s/^{/{\"variant\": \"$v\", /" >> $out
done
done
cat $out

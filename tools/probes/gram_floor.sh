#!/bin/bash
# Same-box floors of the row-space backward (csrc/nmf_cf_gram.hip): the shipped kernel, the same kernel at two waves per SIMD,
# and its memory skeleton (loads, exchanges, stores; -DFZ_PROBE_GRAM_NOMATH).  Build the two alternative libraries first:
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_nomath.so -DFZ_PROBE_GRAM_NOMATH nmf_cf_gram.hip
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_w2.so -DFZ_GRAM_WAVES=2 nmf_cf_gram.hip
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_pad1k.so -DFZ_PROBE_PLANE_PAD=1024 nmf_cf.hip   (planes 4 KiB further apart; pad33: 8448)
#   python tools/probes/build_alt.py tools/probes/bin/lib_gram_pad1k_nomath.so -DFZ_PROBE_PLANE_PAD=1024,-DFZ_PROBE_GRAM_NOMATH nmf_cf.hip nmf_cf_gram.hip
out=${1:-gpurun_out/r05/gram_floor.jsonl}
mkdir -p $(dirname $out); : > $out
for v in ${VARIANTS:-shipped nodma w2 nomath nomath_nodma pad1k pad1k_nomath pad33 general general_pad1k}; do
  case $v in
    shipped) env= ;;
    nodma) env="FZ_CF_GRAM_DMA=0" ;;
    w2) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_w2.so" ;;
    nomath) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_nomath.so" ;;
    nomath_nodma) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_nomath.so FZ_CF_GRAM_DMA=0" ;;
    pad1k) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_pad1k.so PROBE_PLANE_PAD=1024" ;;
    pad1k_nomath) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_pad1k_nomath.so PROBE_PLANE_PAD=1024" ;;
    pad33) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_pad33.so PROBE_PLANE_PAD=8448" ;;
    general) env="FZ_CF_GRAM=0" ;;
    general_pad1k) env="FZ_LIB_PATH=tools/probes/bin/lib_gram_pad1k.so PROBE_PLANE_PAD=1024 FZ_CF_GRAM=0" ;;
  esac
  env $env python tools/probes/gram_bwd.py time 2>/dev/null | grep '"relu_gate": 1' | sed "s/^{/{\"variant\": \"$v\", /" >> $out
done
cat $out

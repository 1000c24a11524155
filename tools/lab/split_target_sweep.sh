for chunk in 128 256 512; do
 for t in 768 512 384 256 160 96; do
  echo "chunk $chunk target $t $(TL_LAB_SPLIT_TARGET=$t python tools/prefill_probe.py --prompt 1024 --chunk $chunk --repeat 3 2>/dev/null | tail -1)"
 done
done

for chunk in 128 256 512 2048; do
  echo "chunk $chunk $(python tools/prefill_probe.py --prompt 2048 --chunk $chunk --repeat 3 2>/dev/null | tail -1)"
done

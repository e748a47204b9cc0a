for a in "resnet50|" "vit_base_patch16_224|" "unicom ViT-L/14|--batch 128" "unicom ViT-L/14|--batch 128 --dtype fp8"; do
  m="${a%%|*}"; e="${a##*|}"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-work --no-roofline --model "$m" $e --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], d['dtype'], d['ms_per_step'], d['value'])"
done

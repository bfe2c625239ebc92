#!/bin/bash
# The ViT / HIPT_4K parity tests under every library switch of INTEGRATION.md section 6 (the rarely-run forms): tools/switch_matrix.sh
for sw in HIPT_GENERIC HIPT_NO_IMG HIPT_NO_PRUNE HIPT_NO_FUSED_ATTN HIPT_NO_PROJ_FOLD HIPT_NO_EMBED_LN; do
  echo "== $sw=1"
  env $sw=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vit256 or hipt4k or vit4k or extract_slide" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | head -12
done

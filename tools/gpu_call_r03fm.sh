#!/bin/bash
if ! timeout 60 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok')"; then
  echo "BAD BOX"; exit 0
fi
timeout 100 python bench.py --only-forward-mode 2>&1 | tail -2 | cut -c1-600

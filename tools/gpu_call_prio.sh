#!/bin/bash
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range()); s=torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0]); print('side', s.priority, 'default', torch.cuda.current_stream().priority)"
bash tools/gpu_ab3.sh 2 PK_SIDE_PRIO=normal PK_SIDE_PRIO=low

#!/bin/bash
# MI355X counterpart of ECAMP/Pre-training/run.sh (same flags; one process per GPU, RCCL over xGMI).
# Run from the repository root.
OMP_NUM_THREADS=1 python -m torch.distributed.run --nproc_per_node=${NGPU:-8} --master-addr 127.0.0.1 -m ecamp_amd.main_pretrain \
    --num_workers 16 \
    --accum_iter 8 \
    --batch_size 256 \
    --model ecamp \
    --norm_pix_loss \
    --mask_ratio 0.75 \
    --epochs 120 \
    --warmup_epochs 40 \
    --lr 1.5e-4 --weight_decay 0.05 \
    --resume ./dataset/mae_vit_base.pth \
    --data_path ./dataset/ \
    --output_dir ../output/ \
    --description "ECAMP pretraining"

#!/bin/bash
# MI355X counterpart of /root/reference ECAMP/Pre-training/run.sh:3-16 -- the same flags and values, line for line; the only changes are
# the launcher (one process per GPU over RCCL/xGMI instead of CUDA_VISIBLE_DEVICES=0,1,2,3 + NCCL) and the module path.
# NGPU defaults to the 8 GPUs of one MI355X node (the reference ran 4); effective batch = NGPU x 256 x 8.
# Run from the repository root:   ./run.sh            (add --synthetic to smoke-test without the MIMIC-CXR files)
OMP_NUM_THREADS=1 python -m torch.distributed.run --nproc_per_node=${NGPU:-8} --master-addr 127.0.0.1 --master_port=${MASTER_PORT:-12345} \
    -m ecamp_amd.main_pretrain \
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
    --description "ECAMP pretraining" "$@"

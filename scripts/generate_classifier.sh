#!/bin/bash
# Classifier generation on the MI355X path with the command line of the reference's scripts/mm_cls/generate_classifier.sh:
#
#   scripts/generate_classifier.sh DATASET SEED SUB_CLASSES N_CTX EVAL_MODE EVAL_TAU GPU_ID
#
# Same positional arguments, same config files (read from a reference checkout: OVMR_REF, default "."), same output directory
# (output_ovmr/generated_classifiers, skipped when it exists), same files in it (mm_classifiers.pt, visual_tokens.pt).  What this path
# needs in addition, from the environment: CLIP_WEIGHTS (OpenAI CLIP .pt: there is no download here) and OVMR_BPE_PATH (default:
# $OVMR_REF/clip/bpe_simple_vocab_16e6.txt.gz).  DATA, MODEL_DIR, CFG, SHOTS, LOADEP, WORKERS may be overridden the same way.
# DRY_RUN=1 prints the command line instead of running it.
# GPU_ID selects the device with HIP_VISIBLE_DEVICES; a comma-separated list starts one rank per GPU (torch.distributed.run, RCCL):
# the classes are sharded over the ranks.
set -e
if [ $# -lt 7 ]; then
    echo "usage: $0 DATASET SEED SUB_CLASSES N_CTX EVAL_MODE EVAL_TAU GPU_ID" >&2
    exit 2
fi
DATASET=$1; SEED=$2; SUB=$3; N_CTX=$4; EVAL_MODE=$5; EVAL_TAU=$6; GPUS=$7
REF=${OVMR_REF:-.}
DATA=${DATA:-./data}
TRAINER=MM_CLS_OP
CFG=${CFG:-vit_b16_c4_ep50_imagenet21k_pretrain}
SHOTS=${SHOTS:-16}
LOADEP=${LOADEP:-30}
MODEL_DIR=${MODEL_DIR:-./checkpoints}
DIR=${DIR:-output_ovmr/generated_classifiers}
: "${CLIP_WEIGHTS:?set CLIP_WEIGHTS to the OpenAI CLIP checkpoint (e.g. ~/.cache/clip/ViT-B-16.pt)}"
export OVMR_BPE_PATH=${OVMR_BPE_PATH:-$REF/clip/bpe_simple_vocab_16e6.txt.gz}
export HIP_VISIBLE_DEVICES=$GPUS
export HSA_ENABLE_IPC_MODE_LEGACY=0
HERE=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$HERE${PYTHONPATH:+:$PYTHONPATH}
N=$(echo "$GPUS" | awk -F, '{print NF}')
ARGS=(--root "$DATA" --seed "$SEED" --trainer $TRAINER
      --dataset-config-file "$REF/configs/datasets/${DATASET}.yaml"
      --config-file "$REF/configs/trainers/${TRAINER}/${CFG}.yaml"
      --output-dir "$DIR" --model-dir "$MODEL_DIR" --load-epoch "$LOADEP"
      --eval_mode "$EVAL_MODE" --eval_tau "$EVAL_TAU" --n_ctx "$N_CTX" --eval-only
      --clip-weights "$CLIP_WEIGHTS" ${WORKERS:+--workers "$WORKERS"}
      DATASET.NUM_SHOTS "$SHOTS" DATASET.SUBSAMPLE_CLASSES "$SUB")
if [ -n "$DRY_RUN" ]; then                     # print the command instead of running it (tests/test_next_rows_cpu.py)
    [ "$N" -gt 1 ] && echo "ranks $N"
    printf '%s\n' python -m ovmr_amd.cli "${ARGS[@]}"
elif [ -d "$DIR" ]; then
    echo "Oops! The results exist at ${DIR} (so skip this job)"
elif [ "$N" -gt 1 ]; then
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29531}" \
        -m ovmr_amd.cli "${ARGS[@]}"
else
    python -m ovmr_amd.cli "${ARGS[@]}"
fi

#!/bin/bash
# Same interface as the reference's Tiny-NewsRec/demo.sh (bash demo.sh train|test|get_teacher_emb), same
# hyper-parameters (demo.sh:3-36); mpirun + horovod are replaced by one process per GPU over RCCL.
n_gpus=${N_GPUS:-4}
mode=$1
exp_name='Tiny-NewsRec-4'
model_dir=../model_all/${exp_name}
common="--model_dir ${model_dir} --npratio 4 --num_words_title 30 --word_embedding_dim 768 --freeze_embedding False \
 --news_dim 256 --save_steps 50000 --max_steps_per_epoch 7000 --apply_bert True --num_attention_heads 16 \
 --num_teacher_layers 12 --num_student_layers 4 --bert_trainable_layer 2 3 --model NAML --model_type tnlrv3 \
 --model_name ../unilmv2/unilm2-base-uncased.bin --config_name ../unilmv2/unilm2-base-uncased-config.json \
 --tokenizer_name ../unilmv2/unilm2-base-uncased-vocab.txt --pooling att --num_teachers 4 \
 --train_data_dir ../MIND/MINDlarge_train --test_data_dir ../MIND/MINDlarge_train"
teachers="--teacher_ckpts ../PLM-NR-12(DP)-1.pt ../PLM-NR-12(DP)-2.pt ../PLM-NR-12(DP)-3.pt ../PLM-NR-12(DP)-4.pt \
 --teacher_emb_paths ../PLM-NR-12(DP)-1.pkl ../PLM-NR-12(DP)-2.pkl ../PLM-NR-12(DP)-3.pkl ../PLM-NR-12(DP)-4.pkl"
if [ "${mode}" == train ]; then
 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${n_gpus} --master-addr 127.0.0.1 --master-port ${PORT:-29511} \
  run.py --mode train --batch_size 32 --epochs 4 --lr 0.0001 --temperature 1.0 --coef 0.2 --user_log_mask False \
  --filename_pat 'behaviors_np4_*.tsv' --use_pretrain_model True --pretrain_model_path ../first_stage_4_layer.pt \
  ${common} ${teachers} ${EXTRA} | tee ../log_all/${exp_name}_train.txt
elif [ "${mode}" == test ]; then
 python -u run.py --mode test --batch_size 128 --user_log_mask True --filename_pat 'behaviors_*.tsv' --load_ckpt_name $2 ${common}
elif [ "${mode}" == get_teacher_emb ]; then
 python -u run.py --mode get_teacher_emb --batch_size 32 --user_log_mask False --num_hidden_layers 12 ${common} ${teachers}
else
 echo "please enter a train or test"
fi

# -*- coding: utf-8 -*-
"""Drop-in for the reference's ``run_train.py`` (reference run_train.py:8-26): ``python run_train.py --yaml
config/train_nisqa_cnn_sa_ap.yaml``.  The YAML keys are the reference's; every batch runs as HIP kernels
(nisqa_amd/train.py), one process per GPU with ``python -m torch.distributed.run`` for data parallel."""
import argparse
import os

import yaml

from nisqa_amd.NISQA_model import nisqaModel

parser = argparse.ArgumentParser()
parser.add_argument('--yaml', required=True, type=str, help='YAML file with config')

if __name__ == "__main__":
    args = vars(parser.parse_args())
    with open(args['yaml'], "r") as ymlfile:
        args_yaml = yaml.load(ymlfile, Loader=yaml.FullLoader)
    args = {**args_yaml, **args}
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl')
    nisqa = nisqaModel(args)
    nisqa.train()

"""CDEvaluator of the reference's models/basic_model.py:9-76 (the inference helper demo.py binds:
CDEvaluator(args).load_checkpoint(name) / .eval() / ._forward_pass(batch) / ._save_predictions()) on the HIP pipelines."""
import os

import torch

from ..misc.imutils import save_image
from .losses import argmax_mask
from .networks import define_G


class CDEvaluator:
    def __init__(self, args):
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        if not (torch.cuda.is_available() and len(args.gpu_ids) > 0):
            raise RuntimeError("dahitra_amd.basic_model.CDEvaluator needs a GPU id (there is no CPU fallback)")
        self.device = torch.device("cuda:%s" % args.gpu_ids[0])
        print(self.device)
        self.checkpoint_dir = args.checkpoint_dir
        self.pred_dir = args.output_folder
        os.makedirs(self.pred_dir, exist_ok=True)
        self.G_pred = None
        self.batch = None
        self.best_val_acc = 0.0
        self.best_epoch_id = 0

    def load_checkpoint(self, checkpoint_name='best_ckpt.pt'):
        path = os.path.join(self.checkpoint_dir, checkpoint_name)
        if not os.path.exists(path):
            raise FileNotFoundError('no such checkpoint %s' % checkpoint_name)
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        sd = checkpoint['model_G_state_dict']
        self.net_G.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})
        self.net_G.to(self.device)
        self.best_val_acc = checkpoint['best_val_acc']
        self.best_epoch_id = checkpoint['best_epoch_id']
        return self.net_G

    def _visualize_pred(self):
        return argmax_mask(self.G_pred).unsqueeze(1) * 255

    def _forward_pass(self, batch):
        self.batch = batch
        img_in1 = batch['A'].to(self.device)
        img_in2 = batch['B'].to(self.device)
        self.shape_h, self.shape_w = img_in1.shape[-2], img_in1.shape[-1]
        with torch.no_grad():
            self.G_pred = self.net_G(img_in1, img_in2)
        return self._visualize_pred()

    def eval(self):
        self.net_G.eval()

    def _save_predictions(self):
        """one binary PNG per sample of the batch (0 / 255), named after the input"""
        preds = self._visualize_pred()
        for i, pred in enumerate(preds):
            file_name = os.path.join(self.pred_dir, self.batch['name'][i].replace('.jpg', '.png'))
            if "." not in os.path.basename(file_name):
                file_name += ".png"
            save_image(pred[0].to(torch.uint8).cpu().numpy(), file_name)

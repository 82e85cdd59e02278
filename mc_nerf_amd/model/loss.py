"""Training loss (reference: model/loss.py:4-58): rgb MSE of the coarse and fine renders plus the
normalised-pixel reprojection MSE of the calibration branch.  A handful of tiny torch ops on [N,3]
and [C,5,2] tensors; it seeds the hand-written backward of RenderTrainFn with 2(rgb-gt)/(3N)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MC_NeRF_Loss(nn.Module):
    def __init__(self, sys_param, tblogger=None):
        super().__init__()
        self.sys_param = sys_param
        self.tblogger = tblogger
        self.global_step = 0
        self.img_h = sys_param["data_img_h"]
        self.img_w = sys_param["data_img_w"]

    def forward(self, loss_dict, epoch_type):
        self.global_step += 1
        if set(loss_dict) == {"intr", "rgb"} and loss_dict["rgb"][0].is_cuda:
            # the NeRF stages: reprojection term (rescaled to value 1 outside the camera-only stage, :20-23) + both rgb terms,
            # value and gradients in one launch (csrc/camera.hip: train_loss_kernel)
            from .render import TrainLossFn
            pd, pt_gt = loss_dict["intr"]
            rgb_c, rgb_f, gt = loss_dict["rgb"]
            return TrainLossFn.apply(pd, pt_gt.to(pd.device), rgb_c, rgb_f, gt, self.img_h, self.img_w, epoch_type != "CAM_PARAM_EPOCH")
        total = 0.0
        if "intr" in loss_dict:
            l_intr = self.get_reproject_loss(loss_dict["intr"])
            # outside the camera-only stage the term is rescaled to value 1 (model/loss.py:20-23)
            total = total + (l_intr if epoch_type == "CAM_PARAM_EPOCH" else l_intr / (l_intr.detach() + 1e-8))
        if "extr" in loss_dict:
            total = total + self.get_reproject_loss(loss_dict["extr"])
        if "rgb" in loss_dict:
            total = total + self.get_rgb_loss(loss_dict["rgb"])
        return total

    def get_rgb_loss(self, rgbs_list):
        rgb_c, rgb_f, gt = rgbs_list
        loss = F.mse_loss(rgb_c, gt)
        if rgb_f is not None:
            loss = loss + F.mse_loss(rgb_f, gt)
        return loss

    def get_reproject_loss(self, rpro_list):
        pd, gt = rpro_list
        if pd.is_cuda:                                   # one fused kernel each way (csrc/camera.hip)
            from .render import ReprojLossFn
            return ReprojLossFn.apply(pd, gt.to(pd.device), self.img_h, self.img_w)
        lx = F.mse_loss(pd[..., 0] / self.img_w, gt[..., 0] / self.img_w)
        ly = F.mse_loss(pd[..., 1] / self.img_h, gt[..., 1] / self.img_h)
        return lx + ly

"""Host mirror of the reference's gradual release (base/parameter_control.py:22-103, ``ResnetParamControl``).

The reference keeps two stacks of PARAMETER-INDEX groups per encoder and, each time it is asked, pops one, flips
``requires_grad`` of ``list(model[modal].parameters())[i]`` for every index of the group, and rebuilds the optimiser
(``trainer.init_optimizer_and_scheduler``).  For the visual encoder the groups are (4..9) = the output layer, (163..186) =
stage 4 and (142..162) = the second half of stage 3 -- always "output layer + a suffix of whole units", which is what
``visual_backbone.IR50`` can train (``_ReleasedHead`` / ``_ReleasedUnit``).  The audio stack of the reference indexes
VGGish parameters 12..17 (its three FC layers); the VGGish mirror has no backward, so releasing it raises.
"""
from operator import itemgetter

import numpy as np


class ResnetParamControl:
    def __init__(self, trainer, gradual_release=1, release_count=8, backbone_mode="ir"):
        self.trainer = trainer
        self.gradual_release = gradual_release
        self.release_count = release_count
        self.backbone_mode = backbone_mode
        self.module_dict = self.init_module_list()
        self.module_stack = self.init_param_group()
        self.early_stop = False

    @staticmethod
    def init_module_list():
        return {"visual": [[(4, 10)], [(163, 187)], [(142, 163)]], "audio": [[(16, 18)], [(14, 16)], [(12, 14)]]}

    def init_param_group(self):
        stack = {"visual": [], "audio": []}
        for modal, ranges in self.module_dict.items():
            for groups in ranges:
                idx = []
                for group in groups:
                    idx += list(np.arange(*group))
                stack[modal].append(idx)
        return stack

    def get_param_group(self, modal):
        return self.module_stack[modal].pop(0)

    def get_current_lr(self):
        return self.trainer.optimizer.param_groups[0]["lr"]

    def release_param(self, model, epoch=0, modalities=("visual",)):
        """``model`` is ``LFAN.spatial`` (a mapping with key 'visual').  Returns the parameters that were released."""
        released = []
        if not self.gradual_release:
            return released
        if self.release_count <= 0:
            print("Early stopped since no further parameters to release!")
            self.early_stop = True
            return released
        for modal in modalities:
            if modal not in model:
                continue
            if modal != "visual":
                raise NotImplementedError("only the visual encoder has a backward on the HIP path (VGGish is frozen)")
            if not self.module_stack[modal]:
                continue
            indices = self.get_param_group(modal)
            params = list(model[modal].parameters())
            for p in list(itemgetter(*indices)(params)):
                p.requires_grad = True
                released.append(p)
        if hasattr(self.trainer, "init_optimizer_and_scheduler"):
            self.trainer.init_optimizer_and_scheduler(epoch=epoch)
        self.release_count -= 1
        if hasattr(self.trainer, "early_stopping"):
            self.trainer.early_stopping_counter = self.trainer.early_stopping
        return released

    def load_trainer(self, trainer):
        self.trainer = trainer

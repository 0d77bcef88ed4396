"""Stand-in for torchlibrosa 0.0.4 ``augmentation`` module (test-only, see package docstring)."""
import torch
import torch.nn as nn


class DropStripes(nn.Module):
    def __init__(self, dim, drop_width, stripes_num):
        super().__init__()
        assert dim in [2, 3]
        self.dim, self.drop_width, self.stripes_num = dim, drop_width, stripes_num

    def forward(self, input):
        assert input.ndimension() == 4
        if self.training is False:
            return input
        batch_size = input.shape[0]
        total_width = input.shape[self.dim]
        for n in range(batch_size):
            self.transform_slice(input[n], total_width)
        return input

    def transform_slice(self, e, total_width):
        for _ in range(self.stripes_num):
            distance = torch.randint(low=0, high=self.drop_width, size=(1,))[0]
            bgn = torch.randint(low=0, high=total_width - distance, size=(1,))[0]
            if self.dim == 2:
                e[:, bgn: bgn + distance, :] = 0
            elif self.dim == 3:
                e[:, :, bgn: bgn + distance] = 0


class SpecAugmentation(nn.Module):
    def __init__(self, time_drop_width, time_stripes_num, freq_drop_width, freq_stripes_num):
        super().__init__()
        self.time_dropper = DropStripes(dim=2, drop_width=time_drop_width, stripes_num=time_stripes_num)
        self.freq_dropper = DropStripes(dim=3, drop_width=freq_drop_width, stripes_num=freq_stripes_num)

    def forward(self, input):
        x = self.time_dropper(input)
        x = self.freq_dropper(x)
        return x

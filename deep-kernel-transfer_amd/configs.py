"""Module-level switches.  The attribute names and values are the reference's (configs.py:1-7) because its callers read them as
`configs.save_dir`, `configs.data_dir[dataset]` and `configs.kernel_type`; nothing else is taken from that file.

kernel_type: 'bncossim' (default), 'cossim', 'linear', 'rbf', 'matern', 'poli1', 'poli2' for classification;
             'rbf' or 'spectral' for regression (the regression drivers fall back to 'rbf' when this names a classification kernel).
"""
kernel_type = 'bncossim'

save_dir = './save/'                     # checkpoints: <save_dir>checkpoints/<dataset>/<model>_<method>[_aug]_<n>way_<k>shot

# file-list roots of the image datasets (only used when such a tree and torchvision are present; `--dataset synthetic` needs none)
data_dir = {name: './filelists/%s/' % name for name in ('CUB', 'miniImagenet', 'omniglot', 'emnist')}

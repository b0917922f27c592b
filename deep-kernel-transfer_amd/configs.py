"""Module-level switches with the reference's names (reference configs.py:1-7)."""
save_dir = './save/'
data_dir = {}
data_dir['CUB'] = './filelists/CUB/'
data_dir['miniImagenet'] = './filelists/miniImagenet/'
data_dir['omniglot'] = './filelists/omniglot/'
data_dir['emnist'] = './filelists/emnist/'
kernel_type = 'bncossim'  # linear, rbf, cossim, bncossim (matern / poli1 / poli2 / spectral: not built yet)

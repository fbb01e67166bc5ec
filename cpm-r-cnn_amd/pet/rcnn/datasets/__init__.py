from .transform import build_transforms
from .dataset import build_dataset, make_train_data_loader, make_test_data_loader
from .evaluation import evaluation, post_processing

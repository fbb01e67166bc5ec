from .transforms import Compose, Resize, RandomHorizontalFlip, ColorJitter, ToTensor, Normalize, DeferredImage

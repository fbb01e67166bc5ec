"""pet.lib.utils.misc.cat (pet/lib/utils/misc.py:99-106), imported by the pooler."""
from pet.rcnn.utils.misc import cat  # noqa: F401

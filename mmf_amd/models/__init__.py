from mmf_amd.models.base_model import BaseModel  # noqa: F401

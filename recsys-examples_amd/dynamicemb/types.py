"""Names of the reference's `dynamicemb.types` that the package exports (types.py:33-116,333-401)."""
from .dynamicemb_config import (BUCKET_ALIGNMENT, DEMB_TABLE_ALIGN_SIZE, EMBEDDING_TYPE, KEY_TYPE,  # noqa: F401
                                MAX_BUCKET_CAPACITY, SCORE_TYPE, DynamicEmbInitializerArgs, DynamicEmbInitializerMode)
from .embedding_admission import AdmissionStrategy, Counter  # noqa: F401

OPT_STATE_TYPE = EMBEDDING_TYPE
COUNTER_TYPE = SCORE_TYPE

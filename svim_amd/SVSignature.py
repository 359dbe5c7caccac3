"""Import-compatible alias of the reference's module name (src/svim/SVSignature.py)."""
from .signatures import (Signature, SignatureDeletion, SignatureInsertion, SignatureInversion,  # noqa: F401
                         SignatureInsertionFrom, SignatureDuplicationTandem, SignatureTranslocation,
                         SignatureClusterUniLocal, SignatureClusterBiLocal)

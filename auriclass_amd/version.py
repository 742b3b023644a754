__package_name__ = "auriclass_amd"
__version__ = "0.1.0"
__reference_version__ = "0.5.4"  # AuriClass release whose behaviour is mirrored
__description__ = ("AuriClass on an MI355X-native MinHash engine: quick estimation of Candida auris clade "
                   "membership, with sketching and distances computed by libmhx on the GPU instead of `mash`")

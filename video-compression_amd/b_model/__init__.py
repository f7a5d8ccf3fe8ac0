"""Drop-in import names of the reference's Flex-Rate package: ``from b_model import b_model``."""

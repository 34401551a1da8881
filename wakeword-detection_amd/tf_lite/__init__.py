"""``tf_lite.*`` import names of the reference's ``utils/tf_lite`` package."""

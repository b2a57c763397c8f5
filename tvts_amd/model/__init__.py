"""Drop-in `model` package of the MI355X TVTSv2 step (model classes, loss, metric) -- see INTEGRATION.md section 2."""

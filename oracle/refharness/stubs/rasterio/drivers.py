def driver_from_extension(path):
    return 'GTiff'
